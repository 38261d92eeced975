// tests/hostemu/hostemu.cpp -- sequential HOST driver of csrc/gsr_math.h (TEST INFRASTRUCTURE).
//
// The authoring container has no GPU.  This harness runs the *same* binary32 per-element arithmetic the
// HIP kernels run (gsr_math.h is __host__ __device__) through a plain sequential pipeline, so the maths can
// be checked against the float64 oracle before a kernel ever touches a GPU.  It is never imported by the
// product; the product path fails loudly when the HIP library is missing.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../3dgs_hierarchical_training_amd/csrc/gsr_math.h"

using namespace gsr;

extern "C" {

struct EmuIn {   // same layout as oracle/gsr_oracle.c:GsrOracleIn
    int32_t N, M, D, W, H, prefiltered;
    float scale_modifier, tanfovx, tanfovy;
    const float *means3D, *scales, *rotations, *cov3D_precomp, *opacities, *shs, *colors_precomp;
    const float *viewmatrix, *projmatrix, *campos, *bg;
};

struct EmuCtx {
    EmuIn in;
    Camera cam;
    std::vector<Splat> splat;
    std::vector<uint32_t> lo;         // sub-ulp remainders of the pixel-space means (gsr_math.h pixel_lo_pack), as the kernels keep them
    std::vector<uint32_t> list;
    std::vector<int64_t> tile_start;
    std::vector<float> finalT, acc;   // acc [P,5]
    std::vector<uint32_t> ncontrib;
    std::vector<float> abs_xy;        // absolute pixel-space means of the override (hostemu_set_absolute_pixels)
};

static Camera make_cam(const EmuIn& in)
{
    Camera c;
    memcpy(c.vm, in.viewmatrix, 64); memcpy(c.pm, in.projmatrix, 64); memcpy(c.cam, in.campos, 12);
    c.tanfovx = in.tanfovx; c.tanfovy = in.tanfovy;
    c.fx = in.W / (2.f * in.tanfovx); c.fy = in.H / (2.f * in.tanfovy);
    c.scale_mod = in.scale_modifier; c.W = in.W; c.H = in.H;
    c.tiles_x = (in.W + kTile - 1) / kTile; c.tiles_y = (in.H + kTile - 1) / kTile;
    c.D = in.D; c.M = in.M;
    return c;
}

// experiment hook: when non-null, overrides the projected state with externally computed (e.g. f64-rounded) values
static const float *g_ov_xy = nullptr, *g_ov_conic = nullptr, *g_ov_rgb = nullptr;
void hostemu_override_geom(const float* xy, const float* conic, const float* rgb) { g_ov_xy = xy; g_ov_conic = conic; g_ov_rgb = rgb; }
// experiment hook: with an override in place, blend with ABSOLUTE binary32 pixel coordinates -- the pixel-space
// mean as given (not re-centred on the image), d = mean - (float)pixel -- which is how the public CUDA module forms its offsets
static int g_abs_pixels = 0;
void hostemu_set_absolute_pixels(int on) { g_abs_pixels = on; }
// experiment hook: 1 = blend without the sub-ulp remainders of the pixel-space means (the arithmetic of rounds 1-4)
static int g_no_lo = 0;
void hostemu_set_no_remainder(int on) { g_no_lo = on; }

EmuCtx* hostemu_forward(const EmuIn* in, float* out_color, float* out_depth, float* out_alpha, int32_t* radii)
{
    EmuCtx* c = new EmuCtx();
    c->in = *in;
    c->cam = make_cam(*in);
    const Camera& cam = c->cam;
    const int N = in->N, W = in->W, H = in->H, T = cam.tiles_x * cam.tiles_y;
    c->splat.resize(N);
    c->lo.assign(N, 0u);
    for (int i = 0; i < N; i++) {
        preprocess_one(cam, in->means3D + 3 * (size_t)i, in->scales ? in->scales + 3 * (size_t)i : nullptr,
                       in->rotations ? in->rotations + 4 * (size_t)i : nullptr,
                       in->cov3D_precomp ? in->cov3D_precomp + 6 * (size_t)i : nullptr, in->opacities[i],
                       in->shs ? in->shs + (size_t)i * in->M * 3 : nullptr, 3, 1,
                       in->colors_precomp ? in->colors_precomp + 3 * (size_t)i : nullptr, c->splat[i], nullptr, false, &c->lo[i]);
        if (g_ov_xy || g_no_lo) c->lo[i] = 0u;      // (an overridden projection has no remainder; g_no_lo: the round-4 arithmetic, for comparison)
        if (g_ov_xy && g_abs_pixels) { if (c->abs_xy.empty()) c->abs_xy.assign(2 * (size_t)N, 0.f); c->abs_xy[2 * i] = g_ov_xy[2 * i]; c->abs_xy[2 * i + 1] = g_ov_xy[2 * i + 1]; }
        if (g_ov_xy && c->splat[i].radius > 0) {
            Splat& s = c->splat[i];
            s.px = g_ov_xy[2 * i] - 0.5f * W; s.py = g_ov_xy[2 * i + 1] - 0.5f * H;
            s.ca = g_ov_conic[3 * i]; s.cb = g_ov_conic[3 * i + 1]; s.cc = g_ov_conic[3 * i + 2];
            s.r = g_ov_rgb[3 * i]; s.g = g_ov_rgb[3 * i + 1]; s.b = g_ov_rgb[3 * i + 2];
        }
        if (radii) radii[i] = c->splat[i].radius;
    }
    // depth order (stable on index), then stable by tile  == (tile, depth bits, index)
    std::vector<uint32_t> order;
    for (int i = 0; i < N; i++) if (c->splat[i].tiles > 0) order.push_back(i);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        uint32_t ka, kb; memcpy(&ka, &c->splat[a].depth, 4); memcpy(&kb, &c->splat[b].depth, 4); return ka < kb; });
    std::vector<std::pair<uint32_t, uint32_t>> inst;  // (tile, gid) in emission order
    for (uint32_t g : order) {
        int x0, y0, x1, y1;
        tile_rect(c->splat[g].px + 0.5f * W, c->splat[g].py + 0.5f * H, c->splat[g].radius, cam.tiles_x, cam.tiles_y, x0, y0, x1, y1);
        const Splat& s = c->splat[g];
        const TileTest tt = make_tile_test(s.px, s.py, s.ca, s.cb, s.cc, s.op);
        uint32_t n = 0;
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++)
                if (tile_accept(tt, x, y, W, H)) { inst.push_back({(uint32_t)(y * cam.tiles_x + x), g}); n++; }
        if (n != s.tiles) abort();   // count (preprocess) and emission must agree
    }
    std::stable_sort(inst.begin(), inst.end(), [](auto& a, auto& b) { return a.first < b.first; });
    c->tile_start.assign(T + 1, 0);
    for (auto& p : inst) c->tile_start[p.first + 1]++;
    for (int t = 0; t < T; t++) c->tile_start[t + 1] += c->tile_start[t];
    c->list.resize(inst.size());
    for (size_t k = 0; k < inst.size(); k++) c->list[k] = inst[k].second;
    const size_t P = (size_t)W * H;
    c->finalT.assign(P, 1.f); c->acc.assign(P * 5, 0.f); c->ncontrib.assign(P, 0);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const int t = (y / kTile) * cam.tiles_x + (x / kTile);
            PixelAcc p = {1.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            uint32_t contributor = 0, last = 0;
            const float ox = (float)((x / kTile) * kTile) - 0.5f * (float)W, oy = (float)((y / kTile) * kTile) - 0.5f * (float)H;   // the tile's first pixel, centred
            const float lx = (float)(x % kTile), ly = (float)(y % kTile);
            for (int64_t k = c->tile_start[t]; k < c->tile_start[t + 1]; k++) {
                contributor++;
                const Splat& s = c->splat[c->list[k]];
                const uint32_t lo = c->lo[c->list[k]];
                float G, dx, dy;
                const float alpha = c->abs_xy.empty()
                    ? pair_alpha_rel(lx, ly, pixel_rel(s.px, pixel_lo_x(lo), ox), pixel_rel(s.py, pixel_lo_y(lo), oy), s.ca, s.cb, s.cc, s.op, G, dx, dy)
                    : pair_alpha((float)x, (float)y, c->abs_xy[2 * (size_t)c->list[k]], c->abs_xy[2 * (size_t)c->list[k] + 1], s.ca, s.cb, s.cc, s.op, G, dx, dy);
                if (alpha == 0.f) continue;
                if (!blend_step_fwd(p, alpha, s.r, s.g, s.b, s.depth)) break;
                last = contributor;
            }
            const size_t pid = (size_t)y * W + x;
            c->finalT[pid] = p.T; c->ncontrib[pid] = last;
            float* a = &c->acc[5 * pid];
            a[0] = p.C0; a[1] = p.C1; a[2] = p.C2; a[3] = p.D; a[4] = p.A;
            if (out_color) for (int ch = 0; ch < 3; ch++) out_color[ch * P + pid] = a[ch] + p.T * in->bg[ch];
            if (out_depth) out_depth[pid] = p.D;
            if (out_alpha) out_alpha[pid] = p.A;
        }
    return c;
}

int64_t hostemu_num_rendered(EmuCtx* c) { return (int64_t)c->list.size(); }

void hostemu_backward(EmuCtx* c, const float* g_color, const float* g_depth, const float* g_alpha, float* d_means3D,
                      float* d_means2D, float* d_opacity, float* d_colors, float* d_shs, float* d_scales,
                      float* d_rotations, float* d_cov3D, float* d_camera /* 35 or null */)
{
    double camacc[35];
    for (int k = 0; k < 35; k++) camacc[k] = 0;
    const EmuIn& in = c->in;
    const Camera& cam = c->cam;
    const int N = in.N, W = in.W, H = in.H;
    const size_t P = (size_t)W * H;
    std::vector<PairGrad> gg(N);
    memset(gg.data(), 0, sizeof(PairGrad) * N);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t pid = (size_t)y * W + x;
            const int t = (y / kTile) * cam.tiles_x + (x / kTile);
            const float* a = &c->acc[5 * pid];
            PixelBwd p;
            p.T = 1.f; p.sC0 = a[0]; p.sC1 = a[1]; p.sC2 = a[2]; p.sD = a[3]; p.sA = a[4];
            p.gC0 = g_color ? g_color[pid] : 0.f; p.gC1 = g_color ? g_color[P + pid] : 0.f; p.gC2 = g_color ? g_color[2 * P + pid] : 0.f;
            p.gD = g_depth ? g_depth[pid] : 0.f; p.gA = g_alpha ? g_alpha[pid] : 0.f;
            p.bgdot = c->finalT[pid] * (in.bg[0] * p.gC0 + in.bg[1] * p.gC1 + in.bg[2] * p.gC2);
            const int64_t s0 = c->tile_start[t];
            const float ox = (float)((x / kTile) * kTile) - 0.5f * (float)W, oy = (float)((y / kTile) * kTile) - 0.5f * (float)H;
            const float lx = (float)(x % kTile), ly = (float)(y % kTile);
            for (uint32_t k = 0; k < c->ncontrib[pid]; k++) {
                const uint32_t g = c->list[s0 + k];
                const Splat& s = c->splat[g];
                float G, dx, dy;
                const float alpha = pair_alpha_rel(lx, ly, pixel_rel(s.px, pixel_lo_x(c->lo[g]), ox), pixel_rel(s.py, pixel_lo_y(c->lo[g]), oy), s.ca, s.cb,
                                                   s.cc, s.op, G, dx, dy);
                if (alpha == 0.f) continue;
                blend_step_bwd(p, alpha, G, dx, dy, s.ca, s.cb, s.cc, s.op, s.r, s.g, s.b, s.depth, gg[g]);
            }
        }
    for (int i = 0; i < N; i++) {
        const Splat& s = c->splat[i];
        float* dsh = d_shs ? d_shs + (size_t)i * in.M * 3 : nullptr;
        for (int k = 0; k < 3; k++) { d_means3D[3 * i + k] = 0.f; d_means2D[3 * i + k] = 0.f; }
        d_opacity[i] = 0.f;
        if (d_colors) for (int k = 0; k < 3; k++) d_colors[3 * i + k] = 0.f;
        if (dsh) for (int k = 0; k < in.M * 3; k++) dsh[k] = 0.f;
        if (d_scales) for (int k = 0; k < 3; k++) d_scales[3 * i + k] = 0.f;
        if (d_rotations) for (int k = 0; k < 4; k++) d_rotations[4 * i + k] = 0.f;
        if (d_cov3D) for (int k = 0; k < 6; k++) d_cov3D[6 * i + k] = 0.f;
        if (s.radius <= 0) continue;
        const PairGrad& g = gg[i];
        GaussGrads o;
        CamGrads cg;
        gauss_backward(cam, in.means3D + 3 * (size_t)i, in.scales ? in.scales + 3 * (size_t)i : nullptr,
                       in.rotations ? in.rotations + 4 * (size_t)i : nullptr,
                       in.cov3D_precomp ? in.cov3D_precomp + 6 * (size_t)i : nullptr, g.gx, g.gy, g.gA, g.gB, g.gC, g.gz, o, &cg);
        float dmean[3] = {o.mean[0], o.mean[1], o.mean[2]};
        const float grgb[3] = {g.gr, g.gg, g.gb};
        if (in.shs) {
            sh_backward(cam, in.means3D + 3 * (size_t)i, in.shs + (size_t)i * in.M * 3, 3, 1, grgb, dsh, 3, 1, dmean);
            for (int k = 0; k < 3; k++) cg.cam[k] = o.mean[k] - dmean[k];
        }
        else for (int k = 0; k < 3; k++) d_colors[3 * i + k] = grgb[k];
        for (int k = 0; k < 16; k++) { camacc[k] += cg.vm[k]; camacc[16 + k] += cg.pm[k]; }
        for (int k = 0; k < 3; k++) camacc[32 + k] += cg.cam[k];
        for (int k = 0; k < 3; k++) d_means3D[3 * i + k] = dmean[k];
        d_means2D[3 * i] = o.mean2d[0]; d_means2D[3 * i + 1] = o.mean2d[1];
        d_opacity[i] = g.gop;
        if (in.cov3D_precomp) { for (int k = 0; k < 6; k++) d_cov3D[6 * i + k] = o.cov[k]; }
        else { for (int k = 0; k < 3; k++) d_scales[3 * i + k] = o.scale[k]; for (int k = 0; k < 4; k++) d_rotations[4 * i + k] = o.rot[k]; }
    }
    if (d_camera) for (int k = 0; k < 35; k++) d_camera[k] = (float)camacc[k];
}

void hostemu_free(EmuCtx* c) { delete c; }

// Property check for tile_rect_tight (gsr_math.h): for `count` pseudo-random splats -- including needle-shaped, faint,
// huge and off-screen ones -- every tile of the reference's 3-sigma rect that passes the exact test must lie inside the
// tight rect.  Returns the number of accepted tiles found OUTSIDE the tight rect (must be 0); *checked = accepted tiles.
long long hostemu_check_tight_rect(int count, unsigned seed, int W, int H, long long* checked)
{
    const int tiles_x = (W + kTile - 1) / kTile, tiles_y = (H + kTile - 1) / kTile;
    unsigned long long st = seed * 6364136223846793005ull + 1442695040888963407ull;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (float)((st >> 40) & 0xffffff) / 16777216.0f; };
    long long bad = 0, seen = 0;
    for (int i = 0; i < count; i++) {
        const float px = (rnd() * 1.4f - 0.2f) * W - 0.5f * W, py = (rnd() * 1.4f - 0.2f) * H - 0.5f * H;   // centred, some off-screen
        const float s1 = expf(rnd() * 7.0f - 1.0f), ratio = expf(rnd() * 4.0f);                         // sigma 0.4 .. 400 px, up to 55:1
        const float s2 = s1 / ratio, th = rnd() * 3.14159265f;
        const float c = cosf(th), sn = sinf(th);
        const float a = c * c * s1 * s1 + sn * sn * s2 * s2 + 0.3f, d = sn * sn * s1 * s1 + c * c * s2 * s2 + 0.3f, b = c * sn * (s1 * s1 - s2 * s2);
        const float det = a * d - b * b;
        const float ca = d / det, cb = -b / det, cc = a / det;   // conic = inverse covariance
        const float mid = 0.5f * (a + d), disc = sqrtf(fmaxf(0.1f, mid * mid - det));
        const int radius = (int)ceilf(3.0f * sqrtf(mid + disc));
        const float r3 = rnd();
        const float op = r3 < 0.2f ? 0.0039f + 0.002f * rnd() : (r3 < 0.6f ? 0.01f + 0.1f * rnd() : rnd());
        int x0, y0, x1, y1, tx0, ty0, tx1, ty1;
        tile_rect(px + 0.5f * W, py + 0.5f * H, radius, tiles_x, tiles_y, x0, y0, x1, y1);
        tile_rect_tight(px, py, radius, ca, cb, cc, op, W, H, tiles_x, tiles_y, tx0, ty0, tx1, ty1);
        const TileTest tt = make_tile_test(px, py, ca, cb, cc, op);
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++)
                if (tile_accept(tt, x, y, W, H)) {
                    seen++;
                    if (!(x >= tx0 && x < tx1 && y >= ty0 && y < ty1)) bad++;
                }
    }
    if (checked) *checked = seen;
    return bad;
}


// Property check for the exact tile culling (gsr_math.h box_accept): whenever ANY pixel centre of a box receives a
// contribution from the splat by the blend's own rule (pair_alpha: power <= 0 and o exp(power) >= 1/255), box_accept must
// say yes -- for tiles (the culling of the binning), 16x8 halves (reach bits of the backward) and arbitrary sub-boxes.
// Returns the number of violations (must be 0); *checked = boxes that had a contributing pixel.
long long hostemu_check_box_accept(int count, unsigned seed, long long* checked)
{
    unsigned long long st = seed * 6364136223846793005ull + 1442695040888963407ull;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (float)((st >> 40) & 0xffffff) / 16777216.0f; };
    long long bad = 0, seen = 0;
    for (int i = 0; i < count; i++) {
        const float px = (rnd() - 0.5f) * 400.f, py = (rnd() - 0.5f) * 400.f;
        const float s1 = expf(rnd() * 5.0f - 1.0f), ratio = expf(rnd() * 3.5f), s2 = s1 / ratio, th = rnd() * 3.14159265f;
        const float c = cosf(th), sn = sinf(th);
        const float a = c * c * s1 * s1 + sn * sn * s2 * s2 + 0.3f, d = sn * sn * s1 * s1 + c * c * s2 * s2 + 0.3f, b = c * sn * (s1 * s1 - s2 * s2);
        const float det = a * d - b * b;
        const float ca = d / det, cb = -b / det, cc = a / det;
        const float r3 = rnd();
        const float op = r3 < 0.3f ? 0.0039f + 0.004f * rnd() : (r3 < 0.6f ? 0.01f + 0.1f * rnd() : rnd());
        const TileTest tt = make_tile_test(px, py, ca, cb, cc, op);
        // a box near the 1/255 contour of the splat (that is where a wrong reject would hide)
        const float reach = sqrtf(fmaxf(0.f, 2.f * logf(255.f * op))) * s1;
        const float ang = rnd() * 6.2831853f, dist = reach * (0.6f + 0.8f * rnd());
        const int bw = 1 + (int)(rnd() * 16.f), bh = 1 + (int)(rnd() * 16.f);
        const float bx0 = floorf(px + dist * cosf(ang)) + 0.5f, by0 = floorf(py + dist * sinf(ang)) + 0.5f;   // pixel centres at k + 0.5
        const float bx1 = bx0 + (float)(bw - 1), by1 = by0 + (float)(bh - 1);
        bool any = false;
        for (int y = 0; y < bh && !any; y++)
            for (int x = 0; x < bw; x++) {
                float G, dx, dy;
                if (pair_alpha(bx0 + (float)x, by0 + (float)y, px, py, ca, cb, cc, op, G, dx, dy) > 0.f) { any = true; break; }
            }
        if (any) {
            seen++;
            if (!box_accept(tt, bx0, by0, bx1, by1)) bad++;
        }
    }
    if (checked) *checked = seen;
    return bad;
}

}
