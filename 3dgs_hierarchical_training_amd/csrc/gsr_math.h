// gsr_math.h -- per-element arithmetic of the rasterizer (binary32), shared by every kernel.
//
// Written from the published 3DGS algorithm (Kerbl et al. 2023 sec. 4-6) and the boundary the
// reference calls (scene/gaussian_model_ht.py:806-894); NOT derived from the un-vendored CUDA
// module.  The functions are __host__ __device__ so the same arithmetic can be driven by a
// sequential host harness (tests/hostemu) in the GPU-less authoring container; the product only
// ever runs them inside the HIP kernels of gsr_kernels.hip.
//
// Conventions
//   * 4x4 matrices are read linearly as column-major (the reference stores them transposed:
//     scene/cameras.py:76-98), i.e. row r of the true matrix is m[r], m[4+r], m[8+r], m[12+r].
//   * quaternion (w,x,y,z) = (r,x,y,z), used un-normalised (the caller normalises:
//     gaussian_model_ht.py:131-133); Sigma = R S^2 R^T (utils/general_utils.py:76-108).
//   * SH layout [M][3] coefficient-major (gaussian_model_ht.py:176-179); basis utils/sh_utils.py:57-100.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define GSR_HD __host__ __device__ __forceinline__
#else
#define GSR_HD inline
#endif

namespace gsr {

constexpr int kTile = 16;
constexpr float kNearZ = 0.2f;
constexpr float kLowpass = 0.3f;
constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kAlphaMax = 0.99f;
constexpr float kTStop = 1e-4f;

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
constexpr float SH_C2_0 = 1.0925484305920792f, SH_C2_1 = -1.0925484305920792f, SH_C2_2 = 0.31539156525252005f,
                SH_C2_3 = -1.0925484305920792f, SH_C2_4 = 0.5462742152960396f;
constexpr float SH_C3_0 = -0.5900435899266435f, SH_C3_1 = 2.890611442640554f, SH_C3_2 = -0.4570457994644658f,
                SH_C3_3 = 0.3731763325901154f, SH_C3_4 = -0.4570457994644658f, SH_C3_5 = 1.445305721320277f,
                SH_C3_6 = -0.5900435899266435f;

// Projected 2D Gaussian, 48 bytes, 16-byte aligned: what the blend kernels gather per instance.
// (Measured in round 2: padding the record to one 64-byte, 64-byte-aligned line -- so that a gather never straddles two
//  lines -- takes the counted HBM traffic of the forward / backward blend from 132 / 203 MB to 118 / 192 MB per launch but
//  leaves their time unchanged (both are VALU-bound) and costs k_preprocess 5 us for the 16 more bytes it writes per
//  Gaussian: kept at 48.)
// px, py are stored RELATIVE TO THE IMAGE CENTRE (W/2, H/2): pixel centres minus the centre are exact in
// binary32, so dx = px - pixel carries only the rounding of |px| <= W/2 instead of W (halves the dominant
// coordinate error at 1920x1080: max image error vs the float64 oracle 1.24e-5 -> below 1e-5).
struct alignas(16) Splat {
    float px, py, ca, cb;      // centred pixel-space mean, conic A, B
    float cc, op, depth, r;    // conic C, opacity, view depth, red
    float g, b;                // green, blue
    int32_t radius;            // ceil(3 sigma) in pixels, 0 = invisible
    uint32_t tiles;            // number of 16x16 tiles that receive a contribution (exact culling)
};
static_assert(sizeof(Splat) == 48, "Splat must be 48 bytes");

// ---- the pixel-space mean beyond binary32 (round 5) ---------------------------------------------------------------------------
// The projection runs in float64 and the mean is rounded ONCE to binary32 -- half an ulp of |px| <= W/2, 1.5e-5 px on a 980-pixel
// frame, 1.2e-4 px 2 056 px from the centre of a 4 112-pixel one.  A sub-pixel splat (conic entries of 1-3 per px^2) turns that into
// a few 1e-5 of its weight two sigma out: the allowance the fuzz cases and the 4112^2 case carried until round 4.  The remainder
// lo = px64 - (double)px is kept as a signed 16-bit count of 2^-16 ulp(px) for each coordinate, in the record's `tiles` word (which no
// kernel reads back: the tile count only lives in the preprocess's registers), and the blends form their offsets as
//     x_rel = fl(fl(px - tile origin) + lo),   dx = x_rel - (pixel's column inside the tile, 0..15)
// -- the first difference is exact or rounded at the ulp of a number of the splat's own size, the second is a difference of nearby
// small numbers.  Same operations, same order in the forward blend, the backward blend and tests/hostemu: their skip decisions agree
// bit for bit, as before.  Cost: a handful of instructions per STAGED instance (not per visit).
GSR_HD int pixel_exp(float p)       // e with p = m 2^e, 0.5 <= |m| < 1 (0 for p = 0)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_frexp_expf(p);
#else
    int e = 0;
    (void)frexpf(p, &e);
    return e;
#endif
}
GSR_HD int pixel_lo_one(double p64, float p)
{
    if (!(fabsf(p) >= 0.0009765625f)) return 0;          // |p| < 2^-10 (or NaN): binary32 resolves 1e-10 px there
    const double q = ldexp(1.0, pixel_exp(p) - 40);        // 2^-16 ulp(p)
    double k = rint((p64 - (double)p) / q);
    k = k < -32768.0 ? -32768.0 : (k > 32767.0 ? 32767.0 : k);
    return (int)k;
}
GSR_HD uint32_t pixel_lo_pack(double px64, float px, double py64, float py)
{
    return ((uint32_t)pixel_lo_one(px64, px) & 0xffffu) | ((uint32_t)pixel_lo_one(py64, py) << 16);
}
// the mean's coordinate relative to `origin` (the centred coordinate of the tile's first pixel column / row), remainder included
GSR_HD float pixel_rel(float p, int lo16, float origin)
{
    const float lo = ldexpf((float)lo16, pixel_exp(p) - 40);
    return (p - origin) + lo;
}
GSR_HD int pixel_lo_x(uint32_t pack) { return (int)(int16_t)(pack & 0xffffu); }
GSR_HD int pixel_lo_y(uint32_t pack) { return (int)(int16_t)(pack >> 16); }

GSR_HD int gsr_popc(uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(v);
#else
    return __builtin_popcount(v);
#endif
}

// What the emission (k_emit) needs of one Gaussian, 8 bytes: the tile rect and -- when the rect has at most 32 tiles
// -- the outcome of the exact tile test as one bit per tile (row-major inside the rect), so that the emission copies
// the decisions k_preprocess already made instead of evaluating the test a second time.
struct alignas(8) TileRec {
    uint32_t mask;   // small rect: bit (ty - y0) * w + (tx - x0) set = tile accepted (count = popcount);
                     // big rect (more than 32 tiles): the NUMBER of accepted tiles, the test is re-run at emission
    uint32_t rect;   // x0 [0:12) | y0 [12:24) | w [24:30) | big [31]
};
static_assert(sizeof(TileRec) == 8, "TileRec must be 8 bytes");
constexpr uint32_t kTileRecBig = 0x80000000u;
constexpr int kTileRecMaskTiles = 32;
constexpr uint32_t kTilesPending = 0xffffffffu;   // Splat::tiles of a large rect whose count the caller still has to fill in
GSR_HD uint32_t tilerec_count(const TileRec& r) { return (r.rect & kTileRecBig) ? r.mask : (uint32_t)gsr_popc(r.mask); }

struct Camera {
    float vm[16], pm[16];
    float cam[3];
    float tanfovx, tanfovy, fx, fy;
    float scale_mod;
    int W, H, tiles_x, tiles_y, D, M;
};

GSR_HD int imin(int a, int b) { return a < b ? a : b; }
GSR_HD int imax(int a, int b) { return a > b ? a : b; }

GSR_HD float fast_rcp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}

GSR_HD float fast_exp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __expf(x);
#else
    return expf(x);
#endif
}

// binary32 view depth with ONE fixed rounding sequence (same in oracle/gsr_oracle.c:depth_key):
// the value is the near-cull test, the sort key and the depth feature.
GSR_HD float depth_key(const float* vm, float x, float y, float z)
{
    return fmaf(vm[10], z, fmaf(vm[6], y, fmaf(vm[2], x, vm[14])));
}

// The forward projection is instantiated in double (RT = double): the per-Gaussian work is negligible next
// to the blend, and a correctly-rounded Splat removes the dominant binary32 error of the image (measured:
// max |err| vs the float64 oracle 9.8e-6 -> 4.1e-6 at 980x545).  The backward uses RT = float.
template <typename RT>
GSR_HD void quat_to_rot(const float q[4], RT R[9])
{
    const RT r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - r * z); R[2] = 2 * (x * z + r * y);
    R[3] = 2 * (x * y + r * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - r * x);
    R[6] = 2 * (x * z - r * y); R[7] = 2 * (y * z + r * x); R[8] = 1 - 2 * (x * x + y * y);
}

// Sigma = L L^T, L = R diag(mod*s); packed xx,xy,xz,yy,yz,zz
template <typename RT>
GSR_HD void cov3d_from_scale_rot(const float s[3], float mod, const float q[4], RT cov[6])
{
    RT R[9];
    quat_to_rot<RT>(q, R);
    const RT s0 = (RT)mod * s[0], s1 = (RT)mod * s[1], s2 = (RT)mod * s[2];
    const RT L00 = R[0] * s0, L01 = R[1] * s1, L02 = R[2] * s2;
    const RT L10 = R[3] * s0, L11 = R[4] * s1, L12 = R[5] * s2;
    const RT L20 = R[6] * s0, L21 = R[7] * s1, L22 = R[8] * s2;
    cov[0] = L00 * L00 + L01 * L01 + L02 * L02;
    cov[1] = L00 * L10 + L01 * L11 + L02 * L12;
    cov[2] = L00 * L20 + L01 * L21 + L02 * L22;
    cov[3] = L10 * L10 + L11 * L11 + L12 * L12;
    cov[4] = L10 * L20 + L11 * L21 + L12 * L22;
    cov[5] = L20 * L20 + L21 * L21 + L22 * L22;
}

// Rows of the 2x3 screen-space Jacobian times the view rotation (M = J * Wr) and the clamped
// view-space position.  Returns false if behind the near plane.
template <typename RT>
struct ProjFrame {
    RT t0, t1, t2;       // view-space mean, x/y after frustum clamp
    RT xmul, ymul;       // 0 where the clamp is active
    RT m0[3], m1[3];     // rows of M
    RT J00, J02, J11, J12;
};

template <typename RT>
GSR_HD void proj_frame(const Camera& c, RT X, RT Y, RT Z, ProjFrame<RT>& f)
{
    const float* vm = c.vm;
    RT t0 = vm[0] * X + vm[4] * Y + vm[8] * Z + vm[12];
    RT t1 = vm[1] * X + vm[5] * Y + vm[9] * Z + vm[13];
    const RT t2 = vm[2] * X + vm[6] * Y + vm[10] * Z + vm[14];
    const RT limx = (RT)1.3 * c.tanfovx, limy = (RT)1.3 * c.tanfovy;
    const RT inv_z = 1 / t2;
    const RT txtz = t0 * inv_z, tytz = t1 * inv_z;
    f.xmul = (txtz < -limx || txtz > limx) ? 0 : 1;
    f.ymul = (tytz < -limy || tytz > limy) ? 0 : 1;
    t0 = (txtz < -limx ? -limx : (txtz > limx ? limx : txtz)) * t2;
    t1 = (tytz < -limy ? -limy : (tytz > limy ? limy : tytz)) * t2;
    f.t0 = t0; f.t1 = t1; f.t2 = t2;
    const RT fx = (RT)c.W / (2 * (RT)c.tanfovx), fy = (RT)c.H / (2 * (RT)c.tanfovy);
    f.J00 = fx * inv_z; f.J02 = -fx * t0 * inv_z * inv_z;
    f.J11 = fy * inv_z; f.J12 = -fy * t1 * inv_z * inv_z;
    for (int k = 0; k < 3; k++) {
        f.m0[k] = f.J00 * vm[k * 4 + 0] + f.J02 * vm[k * 4 + 2];
        f.m1[k] = f.J11 * vm[k * 4 + 1] + f.J12 * vm[k * 4 + 2];
    }
}

// cov2D = M Sigma M^T + 0.3 I  -> (a, b, c)
template <typename RT>
GSR_HD void cov2d_from_frame(const ProjFrame<RT>& f, const RT cov[6], RT& a, RT& b, RT& c, RT Sm0[3], RT Sm1[3])
{
    Sm0[0] = cov[0] * f.m0[0] + cov[1] * f.m0[1] + cov[2] * f.m0[2];
    Sm0[1] = cov[1] * f.m0[0] + cov[3] * f.m0[1] + cov[4] * f.m0[2];
    Sm0[2] = cov[2] * f.m0[0] + cov[4] * f.m0[1] + cov[5] * f.m0[2];
    Sm1[0] = cov[0] * f.m1[0] + cov[1] * f.m1[1] + cov[2] * f.m1[2];
    Sm1[1] = cov[1] * f.m1[0] + cov[3] * f.m1[1] + cov[4] * f.m1[2];
    Sm1[2] = cov[2] * f.m1[0] + cov[4] * f.m1[1] + cov[5] * f.m1[2];
    a = f.m0[0] * Sm0[0] + f.m0[1] * Sm0[1] + f.m0[2] * Sm0[2] + (RT)0.3;
    b = f.m0[0] * Sm1[0] + f.m0[1] * Sm1[1] + f.m0[2] * Sm1[2];
    c = f.m1[0] * Sm1[0] + f.m1[1] * Sm1[1] + f.m1[2] * Sm1[2] + (RT)0.3;
}

GSR_HD void tile_rect(float px, float py, int radius, int tiles_x, int tiles_y, int& x0, int& y0, int& x1, int& y1)
{
    const float inv = 1.0f / kTile;
    const float r = (float)radius;
    x0 = imin(tiles_x, imax(0, (int)((px - r) * inv)));
    y0 = imin(tiles_y, imax(0, (int)((py - r) * inv)));
    x1 = imin(tiles_x, imax(0, (int)((px + r + (kTile - 1)) * inv)));
    y1 = imin(tiles_y, imax(0, (int)((py + r + (kTile - 1)) * inv)));
}

// ------------------------------------------------------------------------------------------------
// Exact, image-preserving tile culling.  A pixel receives a contribution only if power <= 0 and
// o*exp(power) >= 1/255, i.e. q = A dx^2 + 2 B dx dy + C dy^2 <= tau = 2 ln(255 o).  A (Gaussian, tile) instance
// whose minimum q over the tile's pixel-centre box exceeds tau (plus slack for binary32 rounding) blends
// nothing and is dropped before the sort.  The rendered image is unchanged; only lists get shorter.
// The SAME function decides the count (k_preprocess) and the emission (k_emit): results must be bit-identical.
// ------------------------------------------------------------------------------------------------
GSR_HD float splat_tau(float opacity)
{
    return opacity > 0.f ? 2.0f * logf(255.0f * opacity) : -1.0f;
}

// Per-Gaussian constants of the tile test, computed ONCE per Gaussian by both users (k_preprocess's count and
// k_emit's emission).  Everything below is written with explicit fmaf / single operations so that the compiler
// has no freedom to contract differently in the two kernels: the two evaluations must agree bit for bit.
struct TileTest {
    float px, py, ca, cb2, cc, rc, ra, hi;   // hi = tau + slack, or negative "reject all" sentinel
    bool none;                               // opacity below 1/255: contributes nowhere
};
GSR_HD TileTest make_tile_test(float px, float py, float ca, float cb, float cc, float op)
{
    TileTest t;
    const float tau = splat_tau(op);
    const float slack = 1e-3f * (1.0f + fabsf(tau));
    t.px = px; t.py = py; t.ca = ca; t.cb2 = 2.f * cb; t.cc = cc;
    t.rc = cb / cc; t.ra = cb / ca;
    t.hi = tau + slack;
    t.none = tau < -slack;
    return t;
}
GSR_HD float quad_form(const TileTest& t, float x, float y)   // ca x^2 + 2 cb x y + cc y^2
{
    const float u = fmaf(t.ca, x, t.cb2 * y);
    return fmaf(t.cc * y, y, u * x);
}
// conservative test on an arbitrary pixel-centre box [bx0,bx1] x [by0,by1] (inclusive, centred coordinates)
GSR_HD bool box_accept(const TileTest& t, float bx0, float by0, float bx1, float by1)
{
    if (t.none) return false;
    const float dx0 = bx0 - t.px, dx1 = bx1 - t.px, dy0 = by0 - t.py, dy1 = by1 - t.py;
    if (dx0 <= 0.f && dx1 >= 0.f && dy0 <= 0.f && dy1 >= 0.f) return true;   // centre inside the box
    // minimum of the convex form over the box is on its boundary: 1-D minimisation along each edge
    float y = fminf(dy1, fmaxf(dy0, -t.rc * dx0));
    float qmin = quad_form(t, dx0, y);
    y = fminf(dy1, fmaxf(dy0, -t.rc * dx1));
    qmin = fminf(qmin, quad_form(t, dx1, y));
    float x = fminf(dx1, fmaxf(dx0, -t.ra * dy0));
    qmin = fminf(qmin, quad_form(t, x, dy0));
    x = fminf(dx1, fmaxf(dx0, -t.ra * dy1));
    qmin = fminf(qmin, quad_form(t, x, dy1));
    return qmin <= t.hi;
}

// the 16x16 tile (tx, ty) of a W x H image
GSR_HD bool tile_accept(const TileTest& t, int tx, int ty, int W, int H)
{
    const float cx = 0.5f * (float)W, cy = 0.5f * (float)H;
    const float bx0 = (float)(tx * kTile) - cx, by0 = (float)(ty * kTile) - cy;
    const float bx1 = fminf(bx0 + (float)(kTile - 1), (float)(W - 1) - cx), by1 = fminf(by0 + (float)(kTile - 1), (float)(H - 1) - cy);
    return box_accept(t, bx0, by0, bx1, by1);
}

// The 3-sigma tile rect of the reference, shrunk to the axis-aligned bounding box of the region where the Gaussian can
// contribute at all: { q <= tau + slack } is an ellipse with half extents sqrt((tau + slack) cov_xx), sqrt(.. cov_yy)
// (cov = inverse of the conic), which is much smaller than ceil(3 sqrt(lambda_max)) for faint or elongated splats.
// Only tiles inside the intersection can pass the exact test, so the set of accepted tiles is unchanged -- the count
// loop and the per-wave emission of large rects just visit fewer candidates.  Everything is computed from the Splat's
// own binary32 fields with explicit fmaf so that k_preprocess and k_emit derive the same rect bit for bit.
GSR_HD void tile_rect_tight(float px, float py, int radius, float ca, float cb, float cc, float op, int W, int H, int tiles_x,
                            int tiles_y, int& x0, int& y0, int& x1, int& y1)
{
    const float ux = px + 0.5f * (float)W, uy = py + 0.5f * (float)H;   // un-centred pixel coordinates
    tile_rect(ux, uy, radius, tiles_x, tiles_y, x0, y0, x1, y1);
    const float tau = splat_tau(op);
    const float hi = tau + 1e-3f * (1.0f + fabsf(tau));
    const float det = fmaf(ca, cc, -(cb * cb));
    if (!(hi > 0.f) || !(det > 0.f)) return;   // contributes nowhere (the test rejects every tile) / degenerate conic
    const float s = hi / det;
    const float hx = sqrtf(s * cc) * 1.0001f + 1.0f, hy = sqrtf(s * ca) * 1.0001f + 1.0f;   // + a pixel of margin
    const float inv = 1.0f / kTile;
    const int bx0 = (int)floorf((ux - hx) * inv), bx1 = (int)floorf((ux + hx) * inv) + 1;
    const int by0 = (int)floorf((uy - hy) * inv), by1 = (int)floorf((uy + hy) * inv) + 1;
    x0 = imax(x0, imin(bx0, x1)); x1 = imin(x1, imax(bx1, x0));
    y0 = imax(y0, imin(by0, y1)); y1 = imin(y1, imax(by1, y0));
}

GSR_HD uint32_t count_accepted_tiles(float px, float py, float ca, float cb, float cc, float op, int x0, int y0, int x1, int y1,
                                     int W, int H, uint32_t* mask_out = nullptr)
{
    const TileTest t = make_tile_test(px, py, ca, cb, cc, op);
    uint32_t n = 0, mask = 0u;   // an OR-reduction over an affine bit index (NOT a running `bit <<= 1`): the loop stays a plain
                                 // reduction loop and the compiler evaluates two tiles per iteration with packed math
    const int w = x1 - x0;
    for (int ty = y0; ty < y1; ty++) {
        const int row = (ty - y0) * w - x0;
        for (int tx = x0; tx < x1; tx++) {
            const bool ok = tile_accept(t, tx, ty, W, H);
            n += ok ? 1u : 0u;
            mask |= (ok ? 1u : 0u) << ((row + tx) & 31);   // only meaningful for rects of at most kTileRecMaskTiles tiles
        }
    }
    if (mask_out) *mask_out = mask;
    return n;
}

// SH colour (before +0.5 / clamp) for one channel; sh points at coefficient 0 of that channel,
// consecutive coefficients are `stride` floats apart.
// `sh0` (optional) points at coefficient 0 when it is stored apart from the others (split dc / rest storage); the
// coefficients k >= 1 are then still addressed as sh[k * stride].
template <typename RT>
GSR_HD RT sh_channel(int deg, const float* sh, int stride, RT x, RT y, RT z, const float* sh0 = nullptr)
{
    constexpr RT SH_C0 = (RT)0.28209479177387814, SH_C1 = (RT)0.4886025119029199;
    constexpr RT SH_C2_0 = (RT)1.0925484305920792, SH_C2_1 = (RT)-1.0925484305920792, SH_C2_2 = (RT)0.31539156525252005,
                 SH_C2_3 = (RT)-1.0925484305920792, SH_C2_4 = (RT)0.5462742152960396;
    constexpr RT SH_C3_0 = (RT)-0.5900435899266435, SH_C3_1 = (RT)2.890611442640554, SH_C3_2 = (RT)-0.4570457994644658,
                 SH_C3_3 = (RT)0.3731763325901154, SH_C3_4 = (RT)-0.4570457994644658, SH_C3_5 = (RT)1.445305721320277,
                 SH_C3_6 = (RT)-0.5900435899266435;
    RT res = SH_C0 * (sh0 ? sh0[0] : sh[0]);
    if (deg > 0) {
        res = res - SH_C1 * y * sh[1 * stride] + SH_C1 * z * sh[2 * stride] - SH_C1 * x * sh[3 * stride];
        if (deg > 1) {
            const RT xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            res = res + SH_C2_0 * xy * sh[4 * stride] + SH_C2_1 * yz * sh[5 * stride] +
                  SH_C2_2 * (2 * zz - xx - yy) * sh[6 * stride] + SH_C2_3 * xz * sh[7 * stride] +
                  SH_C2_4 * (xx - yy) * sh[8 * stride];
            if (deg > 2) {
                res = res + SH_C3_0 * y * (3 * xx - yy) * sh[9 * stride] + SH_C3_1 * xy * z * sh[10 * stride] +
                      SH_C3_2 * y * (4 * zz - xx - yy) * sh[11 * stride] +
                      SH_C3_3 * z * (2 * zz - 3 * xx - 3 * yy) * sh[12 * stride] +
                      SH_C3_4 * x * (4 * zz - xx - yy) * sh[13 * stride] +
                      SH_C3_5 * z * (xx - yy) * sh[14 * stride] + SH_C3_6 * x * (xx - 3 * yy) * sh[15 * stride];
            }
        }
    }
    return res;
}

// SH colour of one Gaussian seen from the camera centre: max(SH(dir) + 0.5, 0) per channel.
// Colour enters the image linearly, so binary32 is enough here (relative error ~1e-7); only the geometry of
// preprocess_one (conic = inverse of a nearly singular 2x2, pixel position) needs the float64 evaluation.
GSR_HD void splat_sh_color(const Camera& c, const float mean[3], const float* sh, int sh_kstride, int sh_cstride, float col[3],
                           const float* sh0 = nullptr)
{
    float dx = mean[0] - c.cam[0], dy = mean[1] - c.cam[1], dz = mean[2] - c.cam[2];
    const float inv_n = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    dx *= inv_n; dy *= inv_n; dz *= inv_n;
    for (int ch = 0; ch < 3; ch++) {
        const float v = sh_channel<float>(c.D, sh + ch * sh_cstride, sh_kstride, dx, dy, dz, sh0 ? sh0 + ch * sh_cstride : nullptr) + 0.5f;
        col[ch] = v < 0.f ? 0.f : v;
    }
}

// Forward projection of one Gaussian.  `sh` may be null when `color_pre` is given (and vice versa);
// `cov_pre` null means build Sigma from scale/rot.  sh coefficient k of channel ch is at
// sh[k*sh_kstride + ch*sh_cstride] (lets the caller hand either the global [M][3] row or an LDS copy).
GSR_HD void preprocess_one(const Camera& c, const float mean[3], const float* scale, const float* rot,
                           const float* cov_pre, float opacity, const float* sh, int sh_kstride, int sh_cstride,
                           const float* color_pre, Splat& out, TileRec* rec = nullptr, bool defer_big = false, uint32_t* lo_pack = nullptr)
{
    typedef double RT;   // see the note above quat_to_rot
    if (rec) { rec->mask = 0u; rec->rect = 1u << 24; }
    if (lo_pack) *lo_pack = 0u;
    out.px = 0.f; out.py = 0.f; out.ca = 0.f; out.cb = 0.f; out.cc = 0.f; out.op = 0.f; out.depth = 0.f;
    out.r = 0.f; out.g = 0.f; out.b = 0.f; out.radius = 0; out.tiles = 0;
    const float zk = depth_key(c.vm, mean[0], mean[1], mean[2]);
    out.depth = zk;
    if (!(zk > kNearZ)) return;
    const RT X = mean[0], Y = mean[1], Z = mean[2];
    const float* pm = c.pm;
    const RT hx = pm[0] * X + pm[4] * Y + pm[8] * Z + pm[12];
    const RT hy = pm[1] * X + pm[5] * Y + pm[9] * Z + pm[13];
    const RT hw = pm[3] * X + pm[7] * Y + pm[11] * Z + pm[15];
    const RT pw = 1 / (hw + (RT)1e-7);
    RT cov[6];
    if (cov_pre) {
        for (int k = 0; k < 6; k++) cov[k] = cov_pre[k];
    } else {
        cov3d_from_scale_rot<RT>(scale, c.scale_mod, rot, cov);
    }
    ProjFrame<RT> f;
    proj_frame<RT>(c, X, Y, Z, f);
    RT a, b, cc, Sm0[3], Sm1[3];
    cov2d_from_frame<RT>(f, cov, a, b, cc, Sm0, Sm1);
    const RT det = a * cc - b * b;
    if (det == 0) return;
    const RT dinv = 1 / det;
    const RT mid = (RT)0.5 * (a + cc);
    RT disc = mid * mid - det;
    disc = sqrt(disc < (RT)0.1 ? (RT)0.1 : disc);
    const RT l1 = mid + disc, l2 = mid - disc;
    const int radius = (int)ceil(3 * sqrt(l1 > l2 ? l1 : l2));
    // centred pixel coordinates: ((ndc+1) W - 1)/2 - W/2 = (ndc W - 1)/2
    const RT px64 = (hx * pw * c.W - 1) * (RT)0.5, py64 = (hy * pw * c.H - 1) * (RT)0.5;
    const float px = (float)px64;
    const float py = (float)py64;
    int x0, y0, x1, y1;
    tile_rect(px + 0.5f * (float)c.W, py + 0.5f * (float)c.H, radius, c.tiles_x, c.tiles_y, x0, y0, x1, y1);
    if ((x1 - x0) * (y1 - y0) == 0) return;   // the reference's visibility rule: 3-sigma rect touches no tile
    float col[3] = {0.f, 0.f, 0.f};   // neither colour source given: geometry only, the caller adds the colour (k_preprocess)
    if (color_pre) {
        col[0] = color_pre[0]; col[1] = color_pre[1]; col[2] = color_pre[2];
    } else if (sh) {
        splat_sh_color(c, mean, sh, sh_kstride, sh_cstride, col);
    }
    out.px = px; out.py = py;
    if (lo_pack) *lo_pack = pixel_lo_pack(px64, px, py64, py);
    out.ca = (float)(cc * dinv); out.cb = (float)(-b * dinv); out.cc = (float)(a * dinv);
    out.op = opacity;
    out.r = col[0]; out.g = col[1]; out.b = col[2];
    out.radius = radius;
    uint32_t mask = 0u;
    tile_rect_tight(out.px, out.py, radius, out.ca, out.cb, out.cc, out.op, c.W, c.H, c.tiles_x, c.tiles_y, x0, y0, x1, y1);
    const int nt = (x1 - x0) * (y1 - y0);
    if (defer_big && nt > kTileRecMaskTiles) {
        // a rect of hundreds of tiles in ONE lane's loop stalls its whole wave: the caller (k_preprocess) counts these
        // with all 64 lanes instead; kTilesPending marks them
        out.tiles = kTilesPending;
        if (rec) { rec->mask = 0u; rec->rect = (uint32_t)x0 | ((uint32_t)y0 << 12) | kTileRecBig; }
        return;
    }
    out.tiles = nt ? count_accepted_tiles(out.px, out.py, out.ca, out.cb, out.cc, out.op, x0, y0, x1, y1, c.W, c.H, rec ? &mask : nullptr)
                   : 0u;
    if (rec) {   // (tile coordinates fit 12 bits: check_common limits the image to 65535 tiles)
        const bool big = nt > kTileRecMaskTiles;
        rec->mask = big ? out.tiles : mask;
        rec->rect = (uint32_t)x0 | ((uint32_t)y0 << 12) | ((uint32_t)((x1 - x0) & 63) << 24) | (big ? kTileRecBig : 0u);
    }
}

// ------------------------------------------------------------------------------------------------
// Blend: one (pixel, Gaussian) step, front to back.
// ------------------------------------------------------------------------------------------------
struct PixelAcc {
    float T, C0, C1, C2, D, A;
};

// returns alpha (0 if the Gaussian is skipped for this pixel)
GSR_HD float pair_alpha(float pxf, float pyf, float sx, float sy, float ca, float cb, float cc, float op, float& G,
                        float& dx, float& dy)
{
    dx = sx - pxf; dy = sy - pyf;
    const float power = -0.5f * (ca * dx * dx + cc * dy * dy) - cb * dx * dy;
    G = fast_exp(power);
    const float alpha = fminf(kAlphaMax, op * G);
    return (power > 0.f || alpha < kAlphaMin) ? 0.f : alpha;
}

// the same from tile-relative coordinates (pixel_rel): xr / yr = the mean relative to the tile's first pixel, lx / ly = the pixel's
// column / row inside the tile
GSR_HD float pair_alpha_rel(float lx, float ly, float xr, float yr, float ca, float cb, float cc, float op, float& G, float& dx, float& dy)
{
    return pair_alpha(lx, ly, xr, yr, ca, cb, cc, op, G, dx, dy);
}

// forward accumulate; returns false when the pixel terminates (this Gaussian is NOT blended)
GSR_HD bool blend_step_fwd(PixelAcc& p, float alpha, float r, float g, float b, float depth)
{
    const float test_T = p.T * (1.f - alpha);
    if (test_T < kTStop) return false;
    const float w = alpha * p.T;
    p.C0 = fmaf(r, w, p.C0); p.C1 = fmaf(g, w, p.C1); p.C2 = fmaf(b, w, p.C2);
    p.D = fmaf(depth, w, p.D); p.A += w;
    p.T = test_T;
    return true;
}

// Backward replay, front to back.  s* start as the forward totals and shrink to the suffix sums:
//   d out/d alpha_i = c_i T_i - (sum_{k>i} c_k a_k T_k + T_f bg) / (1 - alpha_i)
struct PixelBwd {
    float T, sC0, sC1, sC2, sD, sA;      // running state
    float gC0, gC1, gC2, gD, gA, bgdot;  // upstream gradients of this pixel; bgdot = T_final * <bg, gC>
};

struct PairGrad {   // per-(pixel,Gaussian) contributions, summed over pixels
    float gx, gy;           // d/d pixel-space mean
    float gA, gB, gC;       // d/d conic (true partials)
    float gop;              // d/d opacity
    float gr, gg, gb, gz;   // d/d colour, d/d depth feature
};

GSR_HD void blend_step_bwd(PixelBwd& p, float alpha, float G, float dx, float dy, float ca, float cb, float cc,
                           float op, float r, float g, float b, float depth, PairGrad& o)
{
    const float w = alpha * p.T;
    p.sC0 = fmaf(-r, w, p.sC0); p.sC1 = fmaf(-g, w, p.sC1); p.sC2 = fmaf(-b, w, p.sC2);
    p.sD = fmaf(-depth, w, p.sD); p.sA -= w;
    const float inv1a = fast_rcp(1.f - alpha);
    float dLda = p.gC0 * (r * p.T - p.sC0 * inv1a) + p.gC1 * (g * p.T - p.sC1 * inv1a) + p.gC2 * (b * p.T - p.sC2 * inv1a);
    dLda += p.gD * (depth * p.T - p.sD * inv1a);
    dLda += p.gA * (p.T - p.sA * inv1a);
    dLda -= p.bgdot * inv1a;
    const float dLdpow = G * op * dLda;   // straight-through at the 0.99 clamp
    o.gx += dLdpow * (-ca * dx - cb * dy);
    o.gy += dLdpow * (-cc * dy - cb * dx);
    o.gA += -0.5f * dx * dx * dLdpow;
    o.gB += -dx * dy * dLdpow;
    o.gC += -0.5f * dy * dy * dLdpow;
    o.gop += G * dLda;
    o.gr += w * p.gC0; o.gg += w * p.gC1; o.gb += w * p.gC2; o.gz += w * p.gD;
    p.T *= (1.f - alpha);
}

// ------------------------------------------------------------------------------------------------
// Per-Gaussian backward (everything after the blend gradients have been reduced per Gaussian).
// Inputs: g_px,g_py  dL/d pixel-space mean; gA,gB,gC dL/d conic (true partials);
//         g_rgb[3] dL/d colour; g_z dL/d depth feature.
// ------------------------------------------------------------------------------------------------
struct GaussGrads {
    float mean[3];
    float mean2d[2];   // module convention: d/d ndc = d/d pixel * (W/2, H/2)
    float scale[3], rot[4], cov[6];
};

// accumulates dL/d(mean) through the colour and writes dL/dsh (coefficient k, channel ch at
// dsh[k*dk + ch*dc]).  Coefficients >= (deg+1)^2 are written as zero up to M.
// `sh0` / `dsh0` (optional): where coefficient 0 lives when it is stored apart from the rest (channel ch at sh0[ch*sc], dsh0[ch*dc]).
GSR_HD void sh_backward(const Camera& c, const float mean[3], const float* sh, int sk, int sc, const float g_rgb_in[3],
                        float* dsh, int dk, int dc, float dmean[3], const float* sh0 = nullptr, float* dsh0 = nullptr)
{
    float dx = mean[0] - c.cam[0], dy = mean[1] - c.cam[1], dz = mean[2] - c.cam[2];
    const float inv_n = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    const float x = dx * inv_n, y = dy * inv_n, z = dz * inv_n;
    float gr[3];
    for (int ch = 0; ch < 3; ch++) {
        const float v = sh_channel<float>(c.D, sh + ch * sc, sk, x, y, z, sh0 ? sh0 + ch * sc : nullptr) + 0.5f;
        gr[ch] = v < 0.f ? 0.f : g_rgb_in[ch];
    }
    float basis[16], bx[16], by[16], bz[16];
    for (int k = 0; k < 16; k++) { basis[k] = 0.f; bx[k] = 0.f; by[k] = 0.f; bz[k] = 0.f; }
    basis[0] = SH_C0;
    if (c.D > 0) {
        basis[1] = -SH_C1 * y; by[1] = -SH_C1;
        basis[2] = SH_C1 * z; bz[2] = SH_C1;
        basis[3] = -SH_C1 * x; bx[3] = -SH_C1;
        if (c.D > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            basis[4] = SH_C2_0 * xy; bx[4] = SH_C2_0 * y; by[4] = SH_C2_0 * x;
            basis[5] = SH_C2_1 * yz; by[5] = SH_C2_1 * z; bz[5] = SH_C2_1 * y;
            basis[6] = SH_C2_2 * (2.f * zz - xx - yy); bx[6] = SH_C2_2 * -2.f * x; by[6] = SH_C2_2 * -2.f * y; bz[6] = SH_C2_2 * 4.f * z;
            basis[7] = SH_C2_3 * xz; bx[7] = SH_C2_3 * z; bz[7] = SH_C2_3 * x;
            basis[8] = SH_C2_4 * (xx - yy); bx[8] = SH_C2_4 * 2.f * x; by[8] = SH_C2_4 * -2.f * y;
            if (c.D > 2) {
                basis[9] = SH_C3_0 * y * (3.f * xx - yy); bx[9] = SH_C3_0 * 6.f * xy; by[9] = SH_C3_0 * (3.f * xx - 3.f * yy);
                basis[10] = SH_C3_1 * xy * z; bx[10] = SH_C3_1 * yz; by[10] = SH_C3_1 * xz; bz[10] = SH_C3_1 * xy;
                basis[11] = SH_C3_2 * y * (4.f * zz - xx - yy); bx[11] = SH_C3_2 * -2.f * xy;
                by[11] = SH_C3_2 * (4.f * zz - xx - 3.f * yy); bz[11] = SH_C3_2 * 8.f * yz;
                basis[12] = SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy); bx[12] = SH_C3_3 * -6.f * xz;
                by[12] = SH_C3_3 * -6.f * yz; bz[12] = SH_C3_3 * (6.f * zz - 3.f * xx - 3.f * yy);
                basis[13] = SH_C3_4 * x * (4.f * zz - xx - yy); bx[13] = SH_C3_4 * (4.f * zz - 3.f * xx - yy);
                by[13] = SH_C3_4 * -2.f * xy; bz[13] = SH_C3_4 * 8.f * xz;
                basis[14] = SH_C3_5 * z * (xx - yy); bx[14] = SH_C3_5 * 2.f * xz; by[14] = SH_C3_5 * -2.f * yz; bz[14] = SH_C3_5 * (xx - yy);
                basis[15] = SH_C3_6 * x * (xx - 3.f * yy); bx[15] = SH_C3_6 * (3.f * xx - 3.f * yy); by[15] = SH_C3_6 * -6.f * xy;
            }
        }
    }
    const int nc = (c.D + 1) * (c.D + 1);
    float gdx = 0.f, gdy = 0.f, gdz = 0.f;
    for (int k = 0; k < 16; k++) {
        if (k < nc) {
            for (int ch = 0; ch < 3; ch++) {
                const float s = (k == 0 && sh0) ? sh0[ch * sc] : sh[k * sk + ch * sc];
                if (k == 0 && dsh0) dsh0[ch * dc] = basis[k] * gr[ch];
                else dsh[k * dk + ch * dc] = basis[k] * gr[ch];
                gdx += bx[k] * s * gr[ch]; gdy += by[k] * s * gr[ch]; gdz += bz[k] * s * gr[ch];
            }
        } else if (k < c.M) {
            for (int ch = 0; ch < 3; ch++) dsh[k * dk + ch * dc] = 0.f;
        }
    }
    const float dot = gdx * x + gdy * y + gdz * z;
    dmean[0] += (gdx - x * dot) * inv_n;
    dmean[1] += (gdy - y * dot) * inv_n;
    dmean[2] += (gdz - z * dot) * inv_n;
}

// Camera gradients (north_star: dL/d viewmatrix; extension of SURVEY.md 8f-4 also covers projmatrix and campos).
// Entries follow the linear (column-major) storage of the matrices: vm[k*4 + r] = row r, column k.
struct CamGrads {
    float vm[16], pm[16], cam[3];
};

// RT = arithmetic of the chain.  The conic -> cov2D step cancels catastrophically for needle-shaped splats
// (-C^2 gA + B C gB - B^2 gC over det^2 with det << A C): in binary32 a 20:1 splat loses three digits of dL/dmean, so the
// kernels run the chain in float64 like the forward projection (K9 is HBM-bound; the extra VALU time hides under it).
template <typename RT>
GSR_HD void gauss_backward_t(const Camera& c, const float mean[3], const float* scale, const float* rot,
                             const float* cov_pre, float g_px, float g_py, float gA_, float gB_, float gC_, float g_z_,
                             GaussGrads& o, CamGrads* cg)
{
    const RT gA = gA_, gB = gB_, gC = gC_, g_z = g_z_;
    const RT X = mean[0], Y = mean[1], Z = mean[2];
    RT cov[6];
    if (cov_pre) {
        for (int k = 0; k < 6; k++) cov[k] = cov_pre[k];
    } else {
        cov3d_from_scale_rot<RT>(scale, c.scale_mod, rot, cov);
    }
    ProjFrame<RT> f;
    proj_frame<RT>(c, X, Y, Z, f);
    RT a, b, cc, Sm0[3], Sm1[3];
    cov2d_from_frame<RT>(f, cov, a, b, cc, Sm0, Sm1);
    // conic -> cov2D (guard 1e-7 recalled from the public module)
    const RT det = a * cc - b * b;
    const RT d2i = (RT)1.0 / (det * det + (RT)1e-7);
    const RT ga = d2i * (-cc * cc * gA + b * cc * gB - b * b * gC);
    const RT gb = d2i * ((RT)2. * b * cc * gA - (det + (RT)2. * b * b) * gB + (RT)2. * a * b * gC);
    const RT gc = d2i * (-b * b * gA + a * b * gB - a * a * gC);
    // cov2D -> Sigma (6 unique entries)
    const RT* m0 = f.m0; const RT* m1 = f.m1;
    RT gS[6];
    gS[0] = ga * m0[0] * m0[0] + gb * m0[0] * m1[0] + gc * m1[0] * m1[0];
    gS[3] = ga * m0[1] * m0[1] + gb * m0[1] * m1[1] + gc * m1[1] * m1[1];
    gS[5] = ga * m0[2] * m0[2] + gb * m0[2] * m1[2] + gc * m1[2] * m1[2];
    gS[1] = (RT)2. * ga * m0[0] * m0[1] + gb * (m0[0] * m1[1] + m0[1] * m1[0]) + (RT)2. * gc * m1[0] * m1[1];
    gS[2] = (RT)2. * ga * m0[0] * m0[2] + gb * (m0[0] * m1[2] + m0[2] * m1[0]) + (RT)2. * gc * m1[0] * m1[2];
    gS[4] = (RT)2. * ga * m0[1] * m0[2] + gb * (m0[1] * m1[2] + m0[2] * m1[1]) + (RT)2. * gc * m1[1] * m1[2];
    // cov2D -> M rows -> J -> view-space mean
    RT gm0[3], gm1[3];
    for (int r = 0; r < 3; r++) { gm0[r] = (RT)2. * ga * Sm0[r] + gb * Sm1[r]; gm1[r] = (RT)2. * gc * Sm1[r] + gb * Sm0[r]; }
    const float* vm = c.vm;
    RT gJ00 = (RT)0., gJ02 = (RT)0., gJ11 = (RT)0., gJ12 = (RT)0.;
    for (int k = 0; k < 3; k++) {
        gJ00 += gm0[k] * vm[k * 4 + 0]; gJ02 += gm0[k] * vm[k * 4 + 2];
        gJ11 += gm1[k] * vm[k * 4 + 1]; gJ12 += gm1[k] * vm[k * 4 + 2];
    }
    const RT iz = (RT)1.0 / f.t2, tz2 = iz * iz, tz3 = tz2 * iz;
    const RT gt0 = f.xmul * -c.fx * tz2 * gJ02;
    const RT gt1 = f.ymul * -c.fy * tz2 * gJ12;
    const RT gt2 = -c.fx * tz2 * gJ00 - c.fy * tz2 * gJ11 + (RT)2. * c.fx * f.t0 * tz3 * gJ02 + (RT)2. * c.fy * f.t1 * tz3 * gJ12;
    // screen position -> mean
    const float* pm = c.pm;
    const RT hx = pm[0] * X + pm[4] * Y + pm[8] * Z + pm[12];
    const RT hy = pm[1] * X + pm[5] * Y + pm[9] * Z + pm[13];
    const RT hw = pm[3] * X + pm[7] * Y + pm[11] * Z + pm[15];
    const RT mw = (RT)1.0 / (hw + (RT)1e-7);
    const RT gnx = g_px * (RT)0.5 * c.W, gny = g_py * (RT)0.5 * c.H;
    o.mean2d[0] = (float)gnx; o.mean2d[1] = (float)gny;
    for (int k = 0; k < 3; k++) {
        RT d = vm[k * 4 + 0] * gt0 + vm[k * 4 + 1] * gt1 + vm[k * 4 + 2] * (gt2 + g_z);
        d += (pm[k * 4 + 0] * mw - pm[k * 4 + 3] * hx * mw * mw) * gnx + (pm[k * 4 + 1] * mw - pm[k * 4 + 3] * hy * mw * mw) * gny;
        o.mean[k] = (float)d;
    }
    if (cg) {
        const RT ph[4] = {X, Y, Z, (RT)1.};
        const RT gtv[3] = {gt0, gt1, gt2 + g_z};
        for (int k = 0; k < 4; k++) {
            // view-space position t = V p_h (rows 0..2) and the depth feature (row 2, folded into gtv[2])
            cg->vm[k * 4 + 0] = (float)(gtv[0] * ph[k]); cg->vm[k * 4 + 1] = (float)(gtv[1] * ph[k]); cg->vm[k * 4 + 2] = (float)(gtv[2] * ph[k]);
            cg->vm[k * 4 + 3] = (float)((RT)0.);
            // homogeneous clip position (rows 0, 1, 3 of the full projection; row 2 is unused by the rasterizer)
            cg->pm[k * 4 + 0] = (float)(gnx * mw * ph[k]); cg->pm[k * 4 + 1] = (float)(gny * mw * ph[k]); cg->pm[k * 4 + 2] = (float)((RT)0.);
            cg->pm[k * 4 + 3] = (float)(-(gnx * hx + gny * hy) * mw * mw * ph[k]);
        }
        for (int k = 0; k < 3; k++) {   // rotation block through M = J Wr
            cg->vm[k * 4 + 0] += (float)(gm0[k] * f.J00);
            cg->vm[k * 4 + 1] += (float)(gm1[k] * f.J11);
            cg->vm[k * 4 + 2] += (float)(gm0[k] * f.J02 + gm1[k] * f.J12);
        }
        cg->cam[0] = (RT)0.; cg->cam[1] = (RT)0.; cg->cam[2] = (RT)0.;
    }
    for (int k = 0; k < 6; k++) o.cov[k] = (float)gS[k];
    for (int k = 0; k < 3; k++) o.scale[k] = (RT)0.;
    for (int k = 0; k < 4; k++) o.rot[k] = (RT)0.;
    if (!cov_pre) {
        RT R[9];
        quat_to_rot<RT>(rot, R);
        const RT sv[3] = {c.scale_mod * scale[0], c.scale_mod * scale[1], c.scale_mod * scale[2]};
        const RT Gf[9] = {gS[0], (RT)0.5 * gS[1], (RT)0.5 * gS[2], (RT)0.5 * gS[1], gS[3], (RT)0.5 * gS[4], (RT)0.5 * gS[2], (RT)0.5 * gS[4], gS[5]};
        RT gR[9];
        for (int bcol = 0; bcol < 3; bcol++) {
            RT gs = (RT)0.;
            for (int arow = 0; arow < 3; arow++) {
                RT acc = (RT)0.;
                for (int k = 0; k < 3; k++) acc += Gf[arow * 3 + k] * R[k * 3 + bcol];
                const RT gL = (RT)2. * acc * sv[bcol];   // dL/dL[arow][bcol]
                gs += gL * R[arow * 3 + bcol];
                gR[arow * 3 + bcol] = gL * sv[bcol];
            }
            o.scale[bcol] = (float)(gs * c.scale_mod);
        }
        const RT r = rot[0], x = rot[1], y = rot[2], z = rot[3];
        o.rot[0] = (RT)2. * (-z * gR[1] + y * gR[2] + z * gR[3] - x * gR[5] - y * gR[6] + x * gR[7]);
        o.rot[1] = (RT)2. * (y * gR[1] + z * gR[2] + y * gR[3] - (RT)2. * x * gR[4] - r * gR[5] + z * gR[6] + r * gR[7] - (RT)2. * x * gR[8]);
        o.rot[2] = (RT)2. * (-(RT)2. * y * gR[0] + x * gR[1] + r * gR[2] + x * gR[3] + z * gR[5] - r * gR[6] + z * gR[7] - (RT)2. * y * gR[8]);
        o.rot[3] = (RT)2. * (-(RT)2. * z * gR[0] - r * gR[1] + x * gR[2] + r * gR[3] - (RT)2. * z * gR[4] + y * gR[5] + x * gR[6] + y * gR[7]);
    }
}


GSR_HD void gauss_backward(const Camera& c, const float mean[3], const float* scale, const float* rot,
                           const float* cov_pre, float g_px, float g_py, float gA, float gB, float gC, float g_z,
                           GaussGrads& o, CamGrads* cg = nullptr)
{
    gauss_backward_t<double>(c, mean, scale, rot, cov_pre, g_px, g_py, gA, gB, gC, g_z, o, cg);
}

}  // namespace gsr
