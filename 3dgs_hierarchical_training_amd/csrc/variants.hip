// variants.hip -- the non-default blend kernels, kept for A/B measurements only (gsr_set_option "blend_fwd_ppt" 1..5,
// "blend_bwd_ppt" 1, 3, 4).  Compiled into libgsr_hip.so only with -DGSR_AB_VARIANTS (build.py build(ab_variants=True) /
// GSR_AB_VARIANTS=1); the default library carries the one forward kernel k_blend_fwd_w6 (with and without reach bits) and
// the one backward kernel k_blend_bwd2, so there is a single place where the forward and backward skip decisions live.
// tests/test_gpu_parity.py::test_blend_variants_agree keeps these honest when they are built.
#include "blend_common.h"
#include "variants.h"

namespace gsr {

// ------------------------------------------------------------------------------------------------
// K7: forward blend.  One workgroup per 16x16 tile, PPT pixels per thread (NT = 256/PPT threads).
// Lists are staged NT instances at a time through a double-buffered LDS ring; the next batch's gathers
// are in flight while the current one is composited.  XCD-aware tile mapping: block b runs on XCD b%8, so
// each XCD gets a contiguous band of tiles (neighbouring tiles share splats -> shared L2 lines).
// ------------------------------------------------------------------------------------------------
template <int PPT>
__global__ __launch_bounds__(256 / PPT) void k_blend_fwd(int W, int H, int tiles_x, int T, const uint2* __restrict__ ranges,
                                                         const uint32_t* __restrict__ list, const Splat* __restrict__ splat,
                                                         const float* __restrict__ bg, float* __restrict__ out_color,
                                                         float* __restrict__ out_depth, float* __restrict__ out_alpha,
                                                         float* __restrict__ img, uint32_t* __restrict__ staged)
{
    constexpr int NT = 256 / PPT;
    __shared__ float4 s_a[1][NT], s_b[1][NT], s_c[1][NT];   // single buffer (see k_blend_bwd2): more tiles per CU
    const int tile = xcd_tile(blockIdx.x, T);
    if (tile >= T) return;
    const int tid = threadIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int px = tx * kTile + (tid & 15);
    const int py0 = ty * kTile + (tid >> 4) * PPT;
    const float pxf = (float)px - 0.5f * (float)W;   // centred pixel coordinates (see Splat)
    const float cyf = 0.5f * (float)H;
    const uint2 rg = ranges[tile];
    const int n = (int)(rg.y - rg.x);
    const int nb = (n + NT - 1) / NT;

    PixelAcc acc[PPT];
    uint32_t last[PPT];
    bool done[PPT];
#pragma unroll
    for (int p = 0; p < PPT; p++) {
        acc[p] = {1.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        last[p] = 0;
        done[p] = !(px < W && (py0 + p) < H);
    }
    float4 ra = {0, 0, 0, 0}, rb = ra, rc = ra;
    if (tid < n) {
        const float4* sp = reinterpret_cast<const float4*>(splat + list[rg.x + tid]);
        ra = sp[0]; rb = sp[1]; rc = sp[2];
    }
    int batches = 0;
    for (int b = 0; b < nb; b++) {
        const int buf = 0;
        if (b) __syncthreads();
        s_a[buf][tid] = ra; s_b[buf][tid] = rb; s_c[buf][tid] = rc;
        bool all_done = true;
#pragma unroll
        for (int p = 0; p < PPT; p++) all_done = all_done && done[p];
        if (__syncthreads_and(all_done)) break;
        batches = b + 1;
        const int nxt = (b + 1) * NT + tid;
        if (nxt < n) {
            const float4* sp = reinterpret_cast<const float4*>(splat + list[rg.x + nxt]);
            ra = sp[0]; rb = sp[1]; rc = sp[2];
        }
        if (!all_done) {
            const int cnt = min(NT, n - b * NT);
            for (int j = 0; j < cnt; j++) {
                const float4 A = s_a[buf][j], B = s_b[buf][j], C = s_c[buf][j];
#pragma unroll
                for (int p = 0; p < PPT; p++) {
                    if (done[p]) continue;
                    float G, dx, dy;
                    const float alpha = pair_alpha(pxf, (float)(py0 + p) - cyf, A.x, A.y, A.z, A.w, B.x, B.y, G, dx, dy);
                    if (alpha == 0.f) continue;
                    if (!blend_step_fwd(acc[p], alpha, B.w, C.x, C.y, B.z)) { done[p] = true; continue; }
                    last[p] = (uint32_t)(b * NT + j + 1);
                }
            }
        }
    }
    if (tid == 0) staged[tile * 4] = (uint32_t)min(n, batches * NT);   // instances actually staged (R_eff)
    const size_t P = (size_t)W * H;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
#pragma unroll
    for (int p = 0; p < PPT; p++) {
        const int py = py0 + p;
        if (px < W && py < H) {
            const size_t pid = (size_t)py * W + px;
            const PixelAcc& a = acc[p];
            img[pid] = a.T;
            reinterpret_cast<uint32_t*>(img)[P + pid] = last[p];
            img[2 * P + pid] = a.C0; img[3 * P + pid] = a.C1; img[4 * P + pid] = a.C2;
            img[5 * P + pid] = a.D; img[6 * P + pid] = a.A;
            out_color[pid] = a.C0 + a.T * bg0;
            out_color[P + pid] = a.C1 + a.T * bg1;
            out_color[2 * P + pid] = a.C2 + a.T * bg2;
            out_depth[pid] = a.D;
            out_alpha[pid] = a.A;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K7 (wave-per-sub-tile variant).  One 64-lane wave = one workgroup = one 8x8 pixel block; the four waves of a
// tile are independent: each stages the tile's list itself in batches of 64 (the gathers of the other three hit
// L2), needs no workgroup barrier, and stops as soon as ITS 64 pixels are saturated.  Compared with one
// 256-thread workgroup per tile this removes the barrier stalls (40% of wave time) and the coarse 256-instance
// staging granularity (tiles were staged to 512 instances when ~300 were needed).
// block b: XCD b & 7, slot k = b >> 3; tile = xcd * per + (k >> 2), sub-tile = k & 3 -> a tile's four waves share an XCD.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_blend_fwd_w(int W, int H, int tiles_x, int T, const uint2* __restrict__ ranges,
                                                    const uint32_t* __restrict__ list, const Splat* __restrict__ splat,
                                                    const float* __restrict__ bg, float* __restrict__ out_color,
                                                    float* __restrict__ out_depth, float* __restrict__ out_alpha,
                                                    float* __restrict__ img, uint32_t* __restrict__ staged4, int interleave,
                                                    float* __restrict__ ckpt, int kCkptFirst)
{
    constexpr int NT = 64;
    __shared__ float4 s_a[2][NT], s_b[2][NT];
    __shared__ float2 s_c[2][NT];   // 10 of the record's 12 floats are used: 5120 B per wave = 32 waves per CU
    const int per = (T + 7) >> 3;
    const int kslot = blockIdx.x >> 3;
    // which tile this XCD slot works on: see slot_tile (a tile's four waves share an XCD under every map)
    const int tile = slot_tile(interleave, (int)(blockIdx.x & 7), kslot >> 2, T, tiles_x);
    const int sub = kslot & 3;
    if (tile < 0) return;
    const int lane = threadIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int px = tx * kTile + (sub & 1) * 8 + (lane & 7);
    const int py = ty * kTile + (sub >> 1) * 8 + (lane >> 3);
    const float pxf = (float)px - 0.5f * (float)W, pyf = (float)py - 0.5f * (float)H;
    const uint2 rg = ranges[tile];
    const int n = (int)(rg.y - rg.x);
    const int nb = (n + NT - 1) / NT;
    PixelAcc acc = {1.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint32_t last = 0;
    bool done = !(px < W && py < H);
    float4 ra = {0, 0, 0, 0}, rb = ra, rc = ra;
    // the staged copy carries the conic pre-multiplied for the exponent in base 2:
    //   log2(G) = A' dx^2 + B' dx dy + C' dy^2,  A' = -log2(e)/2 * A, B' = -log2(e) * B, C' = -log2(e)/2 * C
    constexpr float kL2E = 1.4426950408889634f;
    if (lane < n) {
        const float4* sp = reinterpret_cast<const float4*>(splat + list[rg.x + lane]);
        ra = sp[0]; rb = sp[1]; rc = sp[2];
        ra.z *= -0.5f * kL2E; ra.w *= -kL2E; rb.x *= -0.5f * kL2E;
    }
    int batches = 0;
    for (int b = 0; b < nb; b++) {
        const int buf = b & 1;
        if (__all(done)) break;
        if (ckpt && !(b & 1) && (b >> 1) >= kCkptFirst) {   // 128-instance boundary deep in a long list: checkpoint
            float* c = ckpt + ((size_t)(rg.x >> 7) + tile + (b >> 1) - kCkptFirst) * kCkptFloats + sub * 64 + lane;
            c[0] = acc.T; c[256] = acc.C0; c[512] = acc.C1; c[768] = acc.C2; c[1024] = acc.D; c[1280] = acc.A;
        }
        s_a[buf][lane] = ra; s_b[buf][lane] = rb; s_c[buf][lane] = make_float2(rc.x, rc.y);
        __syncthreads();   // single-wave workgroup: just orders the LDS writes before the broadcast reads
        batches = b + 1;
        const int nxt = (b + 1) * NT + lane;
        if (nxt < n) {
            const float4* sp = reinterpret_cast<const float4*>(splat + list[rg.x + nxt]);
            ra = sp[0]; rb = sp[1]; rc = sp[2];
            ra.z *= -0.5f * kL2E; ra.w *= -kL2E; rb.x *= -0.5f * kL2E;
        }
        const int cnt = min(NT, n - b * NT);
        for (int j = 0; j < cnt; j++) {
            const float4 A = s_a[buf][j], B = s_b[buf][j];
            const float2 C = s_c[buf][j];
            if (done) continue;
            const float dx = A.x - pxf, dy = A.y - pyf;
            const float p2 = fmaf(B.x * dy, dy, fmaf(A.w, dy, A.z * dx) * dx);   // log2 of the Gaussian weight
#if defined(__HIP_DEVICE_COMPILE__)
            const float alpha = fminf(kAlphaMax, B.y * __builtin_amdgcn_exp2f(p2));
#else
            const float alpha = fminf(kAlphaMax, B.y * exp2f(p2));
#endif
            if (p2 > 0.f || alpha < kAlphaMin) continue;
            if (!blend_step_fwd(acc, alpha, B.w, C.x, C.y, B.z)) { done = true; continue; }
            last = (uint32_t)(b * NT + j + 1);
        }
    }
    if (lane == 0) staged4[tile * 4 + sub] = (uint32_t)min(n, batches * NT);
    if (px < W && py < H) {
        const size_t P = (size_t)W * H, pid = (size_t)py * W + px;
        img[pid] = acc.T;
        reinterpret_cast<uint32_t*>(img)[P + pid] = last;
        img[2 * P + pid] = acc.C0; img[3 * P + pid] = acc.C1; img[4 * P + pid] = acc.C2;
        img[5 * P + pid] = acc.D; img[6 * P + pid] = acc.A;
        out_color[pid] = acc.C0 + acc.T * bg[0];
        out_color[P + pid] = acc.C1 + acc.T * bg[1];
        out_color[2 * P + pid] = acc.C2 + acc.T * bg[2];
        out_depth[pid] = acc.D;
        out_alpha[pid] = acc.A;
    }
}

// ------------------------------------------------------------------------------------------------
// K7 (packed variant): two vertically adjacent pixels per lane as float2 -> v_pk_* arithmetic, branch-free
// per-pixel skip / stop (masked alpha), only the whole-wave skip is a branch.  Same power expression as
// k_blend_bwd2 so forward and backward agree on every skip decision bit for bit.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_blend_fwd2(int W, int H, int tiles_x, int T, const uint2* __restrict__ ranges,
                                                    const uint32_t* __restrict__ list, const Splat* __restrict__ splat,
                                                    const float* __restrict__ bg, float* __restrict__ out_color,
                                                    float* __restrict__ out_depth, float* __restrict__ out_alpha,
                                                    float* __restrict__ img, uint32_t* __restrict__ staged)
{
    constexpr int NT = 128;
    __shared__ float4 s_a[2][NT], s_b[2][NT], s_c[2][NT];
    const int tile = xcd_tile(blockIdx.x, T);
    if (tile >= T) return;
    const int tid = threadIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int px = tx * kTile + (tid & 15);
    const int py0 = ty * kTile + (tid >> 4) * 2;
    const float pxf = (float)px - 0.5f * (float)W;   // centred pixel coordinates (see Splat)
    const float cyf = 0.5f * (float)H;
    const f2 pyf = {(float)py0 - cyf, (float)(py0 + 1) - cyf};
    const uint2 rg = ranges[tile];
    const int n = (int)(rg.y - rg.x);
    const int nb = (n + NT - 1) / NT;
    f2 Tt = {1.f, 1.f}, C0 = {0.f, 0.f}, C1 = C0, C2 = C0, Dd = C0, Aa = C0;
    uint32_t last0 = 0, last1 = 0;
    bool done0 = !(px < W && py0 < H), done1 = !(px < W && (py0 + 1) < H);
    float4 ra = {0, 0, 0, 0}, rb = ra, rc = ra;
    if (tid < n) {
        const float4* sp = reinterpret_cast<const float4*>(splat + list[rg.x + tid]);
        ra = sp[0]; rb = sp[1]; rc = sp[2];
    }
    int batches = 0;
    for (int b = 0; b < nb; b++) {
        const int buf = b & 1;
        s_a[buf][tid] = ra; s_b[buf][tid] = rb; s_c[buf][tid] = rc;
        const bool all_done = done0 && done1;
        if (__syncthreads_and(all_done)) break;
        batches = b + 1;
        const int nxt = (b + 1) * NT + tid;
        if (nxt < n) {
            const float4* sp = reinterpret_cast<const float4*>(splat + list[rg.x + nxt]);
            ra = sp[0]; rb = sp[1]; rc = sp[2];
        }
        if (!all_done) {
            const int cnt = min(NT, n - b * NT);
            for (int j = 0; j < cnt; j++) {
                const float4 A = s_a[buf][j], B = s_b[buf][j], C = s_c[buf][j];
                const float ca = A.z, cb = A.w, cc = B.x, op = B.y;
                const float dx = A.x - pxf;
                const f2 dy = A.y - pyf;
                const float hx = ca * dx * dx, bx = cb * dx;
                const f2 power = -0.5f * (cc * dy * dy + hx) - bx * dy;
                f2 alpha = {op * fast_exp(power.x), op * fast_exp(power.y)};
                alpha.x = fminf(kAlphaMax, alpha.x); alpha.y = fminf(kAlphaMax, alpha.y);
                const bool v0 = !(power.x > 0.f || alpha.x < kAlphaMin) && !done0;
                const bool v1 = !(power.y > 0.f || alpha.y < kAlphaMin) && !done1;
                if (__any(v0 || v1)) {
                    const f2 test = Tt * (1.f - alpha);
                    const bool stop0 = v0 && test.x < kTStop, stop1 = v1 && test.y < kTStop;
                    const bool b0 = v0 && !stop0, b1 = v1 && !stop1;   // blended
                    done0 = done0 || stop0; done1 = done1 || stop1;
                    alpha.x = b0 ? alpha.x : 0.f; alpha.y = b1 ? alpha.y : 0.f;
                    const f2 w = alpha * Tt;
                    C0 += B.w * w; C1 += C.x * w; C2 += C.y * w; Dd += B.z * w; Aa += w;
                    Tt.x = b0 ? test.x : Tt.x; Tt.y = b1 ? test.y : Tt.y;
                    const uint32_t idx = (uint32_t)(b * NT + j + 1);
                    last0 = b0 ? idx : last0; last1 = b1 ? idx : last1;
                }
            }
        }
    }
    if (tid == 0) staged[tile * 4] = (uint32_t)min(n, batches * NT);
    const size_t P = (size_t)W * H;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int py = py0 + p;
        if (px < W && py < H) {
            const size_t pid = (size_t)py * W + px;
            const float t = Tt[p];
            img[pid] = t;
            reinterpret_cast<uint32_t*>(img)[P + pid] = p ? last1 : last0;
            img[2 * P + pid] = C0[p]; img[3 * P + pid] = C1[p]; img[4 * P + pid] = C2[p];
            img[5 * P + pid] = Dd[p]; img[6 * P + pid] = Aa[p];
            out_color[pid] = C0[p] + t * bg0;
            out_color[P + pid] = C1[p] + t * bg1;
            out_color[2 * P + pid] = C2[p] + t * bg2;
            out_depth[pid] = Dd[p];
            out_alpha[pid] = Aa[p];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K8: backward blend.  Same staging; front-to-back replay from the stored totals.  The per-(pixel,Gaussian)
// contributions are summed over the thread's pixels, reduced across the wave with DPP, combined across
// the tile's waves in LDS, and flushed with ONE set of 10 float atomics per (tile, Gaussian) by the
// thread that staged that Gaussian (parallel across lanes, not serialised on a leader).
// ggrad record (12 floats / Gaussian): gx gy gA gB gC gop gr gg gb gz - -
// ------------------------------------------------------------------------------------------------
template <int PPT>
__global__ __launch_bounds__(256 / PPT) void k_blend_bwd(int W, int H, int tiles_x, int T, const uint2* __restrict__ ranges,
                                                         const uint32_t* __restrict__ list, const Splat* __restrict__ splat,
                                                         const float* __restrict__ bg, const float* __restrict__ img,
                                                         const float* __restrict__ g_color, const float* __restrict__ g_depth,
                                                         const float* __restrict__ g_alpha, float* __restrict__ ggrad)
{
    constexpr int NT = 256 / PPT;
    constexpr int NW = NT / 64;
    __shared__ float4 s_a[2][NT], s_b[2][NT], s_c[2][NT];
    __shared__ uint32_t s_gid[2][NT];
    __shared__ float s_part[NW][NT][10];
    __shared__ uint32_t s_max[NW];
    const int tile = xcd_tile(blockIdx.x, T);
    if (tile >= T) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int px = tx * kTile + (tid & 15);
    const int py0 = ty * kTile + (tid >> 4) * PPT;
    const float pxf = (float)px - 0.5f * (float)W;   // centred pixel coordinates (see Splat)
    const float cyf = 0.5f * (float)H;
    const uint2 rg = ranges[tile];
    const size_t P = (size_t)W * H;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];

    PixelBwd st[PPT];
    uint32_t ncon[PPT];
    uint32_t nmax = 0;
#pragma unroll
    for (int p = 0; p < PPT; p++) {
        const int py = py0 + p;
        ncon[p] = 0;
        st[p] = {1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (px < W && py < H) {
            const size_t pid = (size_t)py * W + px;
            ncon[p] = reinterpret_cast<const uint32_t*>(img)[P + pid];
            st[p].sC0 = img[2 * P + pid]; st[p].sC1 = img[3 * P + pid]; st[p].sC2 = img[4 * P + pid];
            st[p].sD = img[5 * P + pid]; st[p].sA = img[6 * P + pid];
            if (g_color) { st[p].gC0 = g_color[pid]; st[p].gC1 = g_color[P + pid]; st[p].gC2 = g_color[2 * P + pid]; }
            if (g_depth) st[p].gD = g_depth[pid];
            if (g_alpha) st[p].gA = g_alpha[pid];
            st[p].bgdot = img[pid] * (bg0 * st[p].gC0 + bg1 * st[p].gC1 + bg2 * st[p].gC2);
            nmax = max(nmax, ncon[p]);
        }
    }
    // block max of n_contrib = how far the list has to be replayed
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, off, 64));
    if (lane == 0) s_max[wave] = nmax;
    __syncthreads();
    nmax = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) nmax = max(nmax, s_max[w]);
    const int n = (int)nmax;
    const int nb = (n + NT - 1) / NT;

    float4 ra = {0, 0, 0, 0}, rb = ra, rc = ra;
    uint32_t rg_id = 0;
    if (tid < n) {
        rg_id = list[rg.x + tid];
        const float4* sp = reinterpret_cast<const float4*>(splat + rg_id);
        ra = sp[0]; rb = sp[1]; rc = sp[2];
    }
    for (int b = 0; b < nb; b++) {
        const int buf = b & 1;
        s_a[buf][tid] = ra; s_b[buf][tid] = rb; s_c[buf][tid] = rc; s_gid[buf][tid] = rg_id;
#pragma unroll
        for (int w = 0; w < NW; w++)
#pragma unroll
            for (int k = 0; k < 10; k++) s_part[w][tid][k] = 0.f;
        __syncthreads();
        const int nxt = (b + 1) * NT + tid;
        if (nxt < n) {
            rg_id = list[rg.x + nxt];
            const float4* sp = reinterpret_cast<const float4*>(splat + rg_id);
            ra = sp[0]; rb = sp[1]; rc = sp[2];
        }
        const int cnt = min(NT, n - b * NT);
        for (int j = 0; j < cnt; j++) {
            const float4 A = s_a[buf][j], B = s_b[buf][j], C = s_c[buf][j];
            const uint32_t idx = (uint32_t)(b * NT + j + 1);
            PairGrad pg = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            bool touched = false;
#pragma unroll
            for (int p = 0; p < PPT; p++) {
                if (idx > ncon[p]) continue;
                float G, dx, dy;
                const float alpha = pair_alpha(pxf, (float)(py0 + p) - cyf, A.x, A.y, A.z, A.w, B.x, B.y, G, dx, dy);
                if (alpha == 0.f) continue;
                blend_step_bwd(st[p], alpha, G, dx, dy, A.z, A.w, B.x, B.y, B.w, C.x, C.y, B.z, pg);
                touched = true;
            }
            if (__any(touched)) {   // wave-uniform
                float v[10] = {pg.gx, pg.gy, pg.gA, pg.gB, pg.gC, pg.gop, pg.gr, pg.gg, pg.gb, pg.gz};
#pragma unroll
                for (int k = 0; k < 10; k++) v[k] = wave_sum_to_lane63(v[k]);
                if (lane == 63) {
#pragma unroll
                    for (int k = 0; k < 10; k++) s_part[wave][j][k] = v[k];
                }
            }
        }
        __syncthreads();
        if (tid < cnt) {
            float v[10];
#pragma unroll
            for (int k = 0; k < 10; k++) {
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < NW; w++) s += s_part[w][tid][k];
                v[k] = s;
            }
            bool nz = false;
#pragma unroll
            for (int k = 0; k < 10; k++) nz = nz || (v[k] != 0.f);
            if (nz) {
                float* dst = ggrad + (size_t)s_gid[buf][tid] * kGG;
#pragma unroll
                for (int k = 0; k < 10; k++) atomicAdd(dst + k, v[k]);
            }
        }
        // s_part is re-zeroed at the top of the next iteration by the same thread rows, after which a
        // barrier follows; the reads above are per-row (tid) so no extra barrier is needed here.
    }
}

template <int PPT>
static void launch_blend_fwd(int W, int H, int tiles_x, int T, const uint2* ranges, const uint32_t* list, const Splat* splat,
                             const float* bg, float* oc, float* od, float* oa, float* img, uint32_t* staged, hipStream_t st)
{
    const int grid = 8 * ((T + 7) / 8);
    hipLaunchKernelGGL(k_blend_fwd<PPT>, dim3(grid), dim3(256 / PPT), 0, st, W, H, tiles_x, T, ranges, list, splat, bg, oc, od, oa, img, staged);
}
template <int PPT>
static void launch_blend_bwd(int W, int H, int tiles_x, int T, const uint2* ranges, const uint32_t* list, const Splat* splat,
                             const float* bg, const float* img, const float* gc, const float* gd, const float* ga, float* gg,
                             hipStream_t st)
{
    const int grid = 8 * ((T + 7) / 8);
    hipLaunchKernelGGL(k_blend_bwd<PPT>, dim3(grid), dim3(256 / PPT), 0, st, W, H, tiles_x, T, ranges, list, splat, bg, img, gc, gd, ga, gg);
}

bool launch_blend_fwd_variant(int ppt, int W, int H, int tiles_x, int T, const uint2* ranges, const uint32_t* list, const Splat* splat,
                              const float* bg, float* oc, float* od, float* oa, float* img, uint32_t* staged, int tile_map, float* ckpt,
                              int ckpt_first, hipStream_t st)
{
    if (ppt == 5)
        hipLaunchKernelGGL(k_blend_fwd_w, dim3(8 * 4 * slots_per_xcd(tile_map, T, tiles_x)), dim3(64), 0, st, W, H, tiles_x, T, ranges, list, splat, bg,
                           oc, od, oa, img, staged, tile_map, ckpt, ckpt_first);
    else if (ppt == 1) launch_blend_fwd<1>(W, H, tiles_x, T, ranges, list, splat, bg, oc, od, oa, img, staged, st);
    else if (ppt == 2)
        hipLaunchKernelGGL(k_blend_fwd2, dim3(8 * ((T + 7) / 8)), dim3(128), 0, st, W, H, tiles_x, T, ranges, list, splat, bg, oc, od, oa, img, staged);
    else if (ppt == 3) launch_blend_fwd<2>(W, H, tiles_x, T, ranges, list, splat, bg, oc, od, oa, img, staged, st);
    else if (ppt == 4) launch_blend_fwd<4>(W, H, tiles_x, T, ranges, list, splat, bg, oc, od, oa, img, staged, st);
    else return false;
    return true;
}

bool launch_blend_bwd_variant(int ppt, int W, int H, int tiles_x, int T, const uint2* ranges, const uint32_t* list, const Splat* splat,
                              const float* bg, const float* img, const float* gc, const float* gd, const float* ga, float* gg, hipStream_t st)
{
    if (ppt == 1) launch_blend_bwd<1>(W, H, tiles_x, T, ranges, list, splat, bg, img, gc, gd, ga, gg, st);
    else if (ppt == 3) launch_blend_bwd<2>(W, H, tiles_x, T, ranges, list, splat, bg, img, gc, gd, ga, gg, st);
    else if (ppt == 4) launch_blend_bwd<4>(W, H, tiles_x, T, ranges, list, splat, bg, img, gc, gd, ga, gg, st);
    else return false;
    return true;
}

}  // namespace gsr
