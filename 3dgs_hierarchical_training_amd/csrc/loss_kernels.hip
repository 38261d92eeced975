// loss_kernels.hip -- fused photometric loss (1-l)*L1 + l*(1-SSIM) on a [3,H,W] render, forward + backward.
//
// "Next" row f-3 of SURVEY.md section 8: the reference computes this loss with five 11x11 depthwise
// convolutions per call in torch (/root/reference/trainer/losses.py:98-136 Loss.forward, :147-209 gaussian /
// create_window / _ssim; lambda_dssim = 0.2 /root/reference/arguments/__init__.py:134), on the clamped render
// (scene/gaussian_model_ht.py:883).  On MI355X those generic convolutions were 39% of the train step
// (profiles/r01_a_bench_kernel_stats.md).  Here one kernel stages a 26x26 halo tile of render and target in
// LDS, runs the separable 11-tap Gaussian for the five moments, evaluates SSIM and its three partial-derivative
// maps, and reduces L1 and SSIM sums; the backward convolves the derivative maps once more (the window is
// symmetric and zero-padded, so the adjoint is the same convolution).  clamp(0,1) of the render is fused in.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gsr.h"

namespace gsr {

constexpr int kLB = 16;           // granularity of the workspace sizing (gsr_loss_workspace_bytes: an upper bound of the blocks)
constexpr int kHalo = 5;          // window 11
constexpr int kTW = 32;           // output tile of one 256-thread workgroup: 32 x (8 RP), RP = output rows per thread in the column pass
constexpr int kIW = kTW + 2 * kHalo;   // 42 staged input columns
#ifndef GSR_LOSS_RP
#define GSR_LOSS_RP 2
#endif
constexpr int kRP = GSR_LOSS_RP, kTH = 8 * kRP, kIH = kTH + 2 * kHalo;
constexpr int kLoadRounds = (kIW * kIH + 255) / 256;
constexpr float kC1 = 0.01f * 0.01f, kC2 = 0.03f * 0.03f;

// exp(-(i-5)^2 / (2*1.5^2)) normalised, i = 0..10 (losses.py:147-150)
__device__ __constant__ float kGauss[11] = {0.0010283801f, 0.0075987581f, 0.0360007721f, 0.1093606895f, 0.2130055377f,
                                            0.2660117249f, 0.2130055377f, 0.1093606895f, 0.0360007721f, 0.0075987581f,
                                            0.0010283801f};

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

// Two moments per register pair: (x, y) and (x^2, y^2) ride through the separable window as float2, so a tap is one
// v_pk_fma_f32 instead of two fmas (per-component fma: same rounding as scalar code).
typedef float lf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ lf2 lfma2(float w, lf2 v, lf2 acc) { return __builtin_elementwise_fma(lf2{w, w}, v, acc); }

// LDS row strides in 4-byte words, all = 4 (mod 32): consecutive lanes work on consecutive ROWS with 16-byte accesses, so
// eight lanes tile the 32 banks exactly (and every 4-column group stays 16-byte aligned)
constexpr int kStrIn2 = 100;      // rows of (x, y) pairs: 42 pairs = 84 words
constexpr int kStrIn1 = 68;       // rows of 42 scalars
constexpr int kStrH2 = 68;        // rows of 32 row-filtered pairs = 64 words
constexpr int kStrH1 = 36;        // rows of 32 row-filtered scalars

// Which tile a workgroup takes.  The hardware deals workgroups to the eight XCDs round-robin in launch order, so with tile = launch
// index every XCD's L2 sees every eighth tile of the frame and fetches each tile's 5-pixel halo for itself (counted: 2.1 x the
// image through the fabric).  Here XCD x takes the x-th contiguous eighth of the (channel, row, column) order instead: a tile's
// neighbours left and right and in the rows above and below run on the same L2.  GSR_LOSS_XCD_MAP=0: tile = launch index.
#ifndef GSR_LOSS_XCD_MAP
#define GSR_LOSS_XCD_MAP 1
#endif
struct LossTile { int bx, by, c; };
__device__ __forceinline__ LossTile loss_tile()
{
#if GSR_LOSS_XCD_MAP
    const uint32_t gx = gridDim.x, gy = gridDim.y, n = gx * gy * gridDim.z;
    const uint32_t lin = (blockIdx.z * gy + blockIdx.y) * gx + blockIdx.x;
    const uint32_t q = n >> 3, r = n & 7u, x = lin & 7u, k = lin >> 3;
    const uint32_t t = (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + k;
    const uint32_t row = t / gx;
    return LossTile{(int)(t - row * gx), (int)(row % gy), (int)(row / gy)};
#else
    return LossTile{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
#endif
}

// Both kernels are bound by VALU issue (a few hundred instructions per pixel, next to 40 bytes of traffic), so what is
// minimised is instructions per output: 32x32 tiles (1.7 staged inputs per output instead of 2.6 at 16x16) and a sliding
// window -- a thread produces FOUR adjacent outputs of a filter pass from 14 inputs it loads once (3.5 LDS reads per
// output instead of 11), and squares / products are formed once per input instead of once per tap.  Every output is still
// the same chain acc = fma(w[k], v[k], acc), k = 0..10, as the 16x16 kernels of round 1 computed.
// Measured at 3 x 545 x 980 (rocprofv3, tools/loss_ab.sh): forward 27.3 -> 21.4 us, backward 23.4 -> 17.2 us with 32x16 tiles
// (RP = 2; 32x32 tiles, RP = 4: 21.5 / 18.4 us -- fewer instructions but 3 instead of 5 workgroups per CU).  Issuing all of a
// thread's global loads before its first LDS store was worth as much as the instruction count (27.5 -> 21.4 us).
// grid (ceil(W/32), ceil(H/32), C); block 256.  maps: [3][C][H][W] = dS/dmu_x, dS/dE[x^2], dS/dE[xy]
__global__ __launch_bounds__(256) void k_loss_fwd(const float* __restrict__ raw, const float* __restrict__ gt, int H, int W,
                                                  int do_clamp, float* __restrict__ maps, float* __restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) float s_in[kIH * kStrIn2];   // (x, y) interleaved
    __shared__ __attribute__((aligned(16))) float s_hA[kIH * kStrH2];    // row-filtered (x, y)
    __shared__ __attribute__((aligned(16))) float s_hB[kIH * kStrH2];    // row-filtered (x^2, y^2)
    __shared__ __attribute__((aligned(16))) float s_hC[kIH * kStrH1];    // row-filtered x y
    __shared__ float s_red[2][4];
    const int tid = threadIdx.x;
    const LossTile lt = loss_tile();
    const int c = lt.c;
    const size_t P = (size_t)H * W;
    const float* x = raw + (size_t)c * P;
    const float* y = gt + (size_t)c * P;
    const int ox = lt.bx * kTW - kHalo, oy = lt.by * kTH - kHalo;
    {   // all loads of the thread are issued before the first LDS store (their latency is paid once, not per round)
        float xv[kLoadRounds], yv[kLoadRounds];
#pragma unroll
        for (int q = 0; q < kLoadRounds; q++) {
            const int i = tid + 256 * q, iy = i / kIW, ix = i - iy * kIW;
            const int gy = oy + iy, gx = ox + ix;
            const bool in = i < kIW * kIH && gx >= 0 && gx < W && gy >= 0 && gy < H;
            const size_t o = in ? (size_t)gy * W + gx : 0;
            xv[q] = in ? x[o] : 0.f; yv[q] = in ? y[o] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < kLoadRounds; q++) {
            const int i = tid + 256 * q, iy = i / kIW, ix = i - iy * kIW;
            if (i < kIW * kIH) *reinterpret_cast<lf2*>(&s_in[iy * kStrIn2 + 2 * ix]) = lf2{do_clamp ? clamp01(xv[q]) : xv[q], yv[q]};
        }
    }
    __syncthreads();
    // row pass: task = (row r, group g of 4 output columns); lanes run over rows
    for (int t = tid; t < kIH * (kTW / 4); t += 256) {
        const int g = t / kIH, r = t - g * kIH;
        lf2 v[14];
        const float4* src = reinterpret_cast<const float4*>(&s_in[r * kStrIn2 + 8 * g]);
#pragma unroll
        for (int q = 0; q < 7; q++) {
            const float4 f = src[q];
            v[2 * q] = lf2{f.x, f.y}; v[2 * q + 1] = lf2{f.z, f.w};
        }
        lf2 sq[14];
        float xy[14];
#pragma unroll
        for (int k = 0; k < 14; k++) { sq[k] = v[k] * v[k]; xy[k] = v[k].x * v[k].y; }
        lf2 hA[4], hB[4];
        float hC[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            hA[i] = lf2{0.f, 0.f}; hB[i] = lf2{0.f, 0.f}; hC[i] = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const float w = kGauss[k];
                hA[i] = lfma2(w, v[i + k], hA[i]); hB[i] = lfma2(w, sq[i + k], hB[i]); hC[i] = fmaf(w, xy[i + k], hC[i]);
            }
        }
        float4* dA = reinterpret_cast<float4*>(&s_hA[r * kStrH2 + 8 * g]);
        float4* dB = reinterpret_cast<float4*>(&s_hB[r * kStrH2 + 8 * g]);
        dA[0] = make_float4(hA[0].x, hA[0].y, hA[1].x, hA[1].y); dA[1] = make_float4(hA[2].x, hA[2].y, hA[3].x, hA[3].y);
        dB[0] = make_float4(hB[0].x, hB[0].y, hB[1].x, hB[1].y); dB[1] = make_float4(hB[2].x, hB[2].y, hB[3].x, hB[3].y);
        *reinterpret_cast<float4*>(&s_hC[r * kStrH1 + 4 * g]) = make_float4(hC[0], hC[1], hC[2], hC[3]);
    }
    __syncthreads();
    // column pass: thread = (column cc, group of kRP output rows); lanes run over columns
    const int cc = tid & (kTW - 1), r0 = (tid / kTW) * kRP;
    lf2 mA[kRP], mB[kRP];
    float mC[kRP];
#pragma unroll
    for (int i = 0; i < kRP; i++) { mA[i] = lf2{0.f, 0.f}; mB[i] = lf2{0.f, 0.f}; mC[i] = 0.f; }
    {
        lf2 a[kRP + 10], b[kRP + 10];
        float d[kRP + 10];
#pragma unroll
        for (int k = 0; k < kRP + 10; k++) {
            a[k] = *reinterpret_cast<const lf2*>(&s_hA[(r0 + k) * kStrH2 + 2 * cc]);
            b[k] = *reinterpret_cast<const lf2*>(&s_hB[(r0 + k) * kStrH2 + 2 * cc]);
            d[k] = s_hC[(r0 + k) * kStrH1 + cc];
        }
#pragma unroll
        for (int i = 0; i < kRP; i++)
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const float w = kGauss[k];
                mA[i] = lfma2(w, a[i + k], mA[i]); mB[i] = lfma2(w, b[i + k], mB[i]); mC[i] = fmaf(w, d[i + k], mC[i]);
            }
    }
    const int gx = lt.bx * kTW + cc;
    const size_t CP = (size_t)gridDim.z * P;
    float ssim = 0.f, l1 = 0.f;
#pragma unroll
    for (int i = 0; i < kRP; i++) {
        const int gy = lt.by * kTH + r0 + i;
        if (gx < W && gy < H) {
            const float mu1 = mA[i].x, mu2 = mA[i].y, e11 = mB[i].x, e22 = mB[i].y, e12 = mC[i];
            const float s11 = e11 - mu1 * mu1, s22 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
            const float A = 2.f * mu1 * mu2 + kC1, B = 2.f * s12 + kC2;
            const float Cc = mu1 * mu1 + mu2 * mu2 + kC1, D = s11 + s22 + kC2;
            const float iCD = 1.f / (Cc * D);
            const float sv = A * B * iCD;
            ssim += sv;
            // partial derivatives of S w.r.t. the three moments that depend on x (mu_x, E[x^2], E[xy])
            const float dS_dmu = 2.f * mu2 * (B - A) * iCD - 2.f * mu1 * sv / Cc + 2.f * mu1 * sv / D;
            const float dS_de11 = -sv / D;
            const float dS_de12 = 2.f * A * iCD;
            const size_t pid = (size_t)c * P + (size_t)gy * W + gx;
            maps[pid] = dS_dmu; maps[CP + pid] = dS_de11; maps[2 * CP + pid] = dS_de12;
            const lf2 ctr = *reinterpret_cast<const lf2*>(&s_in[(r0 + i + kHalo) * kStrIn2 + 2 * (cc + kHalo)]);
            l1 += fabsf(ctr.x - ctr.y);
        }
    }
    // block reduction (wave shuffles, then 4 partials)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { ssim += __shfl_xor(ssim, off, 64); l1 += __shfl_xor(l1, off, 64); }
    if ((tid & 63) == 0) { s_red[0][tid >> 6] = ssim; s_red[1][tid >> 6] = l1; }
    __syncthreads();
    if (tid == 0) {
        const size_t b = ((size_t)lt.c * gridDim.y + lt.by) * gridDim.x + lt.bx;   // the TILE's slot: the finishing sum does not see the map
        partial[2 * b] = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
        partial[2 * b + 1] = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
    }
}

// single block: deterministic reduction of the per-block partials -> out[0] = loss, out[1] = mean ssim, out[2] = mean l1
// images > 1 (a stack of independent images, gsr_loss_forward_batched): inv_count is ONE image's 1 / (C H W), so the two sums are
// sums of per-image means; out[0] = the SUM of the images' losses, out[1] / out[2] their mean SSIM / mean L1
__global__ __launch_bounds__(1024) void k_loss_finish(const float* __restrict__ partial, int nblocks, float inv_count, float lambda,
                                                      float* __restrict__ out, int images, int nout = 3, float* __restrict__ loss_dup = nullptr)
{
    __shared__ double s0[1024], s1[1024];
    double a = 0, b = 0;
    for (int i = threadIdx.x; i < nblocks; i += 1024) { a += partial[2 * i]; b += partial[2 * i + 1]; }
    s0[threadIdx.x] = a; s1[threadIdx.x] = b;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (threadIdx.x < off) { s0[threadIdx.x] += s0[threadIdx.x + off]; s1[threadIdx.x] += s1[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double ms = s0[0] * inv_count, ml = s1[0] * inv_count;
        out[0] = (float)((1.0 - lambda) * ml + lambda * ((double)images - ms));
        out[1] = (float)(ms / images); out[2] = (float)(ml / images);
        if (nout >= 6) {   // what Loss.forward reports beside the loss (/root/reference/trainer/losses.py:128-136), so the caller needs no torch ops for them
            out[3] = (float)((1.0 - lambda) * (ml / images));    // loss_rgb = (1 - lambda) * mean L1
            out[4] = (float)(1.0 - ms / images);                 // loss_dssim = 1 - mean SSIM
            out[5] = 0.f;                                        // loss_depth when no depth term is configured
        }
        if (loss_dup) *loss_dup = out[0];
    }
}

__global__ __launch_bounds__(256) void k_loss_bwd(const float* __restrict__ raw, const float* __restrict__ gt, int H, int W,
                                                  int do_clamp, const float* __restrict__ maps, const float* __restrict__ gscale,
                                                  float inv_count, float lambda, float* __restrict__ d_raw)
{
    __shared__ __attribute__((aligned(16))) float s_mA[kIH * kStrIn2];   // (dS/dmu_x, dS/dE[x^2])
    __shared__ __attribute__((aligned(16))) float s_mC[kIH * kStrIn1];   // dS/dE[xy]
    __shared__ __attribute__((aligned(16))) float s_hA[kIH * kStrH2];
    __shared__ __attribute__((aligned(16))) float s_hC[kIH * kStrH1];
    const int tid = threadIdx.x;
    const LossTile lt = loss_tile();
    const int c = lt.c;
    const size_t P = (size_t)H * W, CP = (size_t)gridDim.z * P;
    const int ox = lt.bx * kTW - kHalo, oy = lt.by * kTH - kHalo;
    {
        float a[kLoadRounds], bb[kLoadRounds], d[kLoadRounds];
#pragma unroll
        for (int q = 0; q < kLoadRounds; q++) {
            const int i = tid + 256 * q, iy = i / kIW, ix = i - iy * kIW;
            const int gy = oy + iy, gx = ox + ix;
            const bool in = i < kIW * kIH && gx >= 0 && gx < W && gy >= 0 && gy < H;
            const size_t pid = in ? (size_t)c * P + (size_t)gy * W + gx : 0;
            a[q] = in ? maps[pid] : 0.f; bb[q] = in ? maps[CP + pid] : 0.f; d[q] = in ? maps[2 * CP + pid] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < kLoadRounds; q++) {
            const int i = tid + 256 * q, iy = i / kIW, ix = i - iy * kIW;
            if (i < kIW * kIH) {
                *reinterpret_cast<lf2*>(&s_mA[iy * kStrIn2 + 2 * ix]) = lf2{a[q], bb[q]};
                s_mC[iy * kStrIn1 + ix] = d[q];
            }
        }
    }
    __syncthreads();
    for (int t = tid; t < kIH * (kTW / 4); t += 256) {
        const int g = t / kIH, r = t - g * kIH;
        lf2 v[14];
        float d[16];
        const float4* srcA = reinterpret_cast<const float4*>(&s_mA[r * kStrIn2 + 8 * g]);
        const float4* srcC = reinterpret_cast<const float4*>(&s_mC[r * kStrIn1 + 4 * g]);
#pragma unroll
        for (int q = 0; q < 7; q++) {
            const float4 f = srcA[q];
            v[2 * q] = lf2{f.x, f.y}; v[2 * q + 1] = lf2{f.z, f.w};
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {   // 16 scalars: the last two lie in the row's padding (stride 68 >= 4 * 7 + 16)
            const float4 f = srcC[q];
            d[4 * q] = f.x; d[4 * q + 1] = f.y; d[4 * q + 2] = f.z; d[4 * q + 3] = f.w;
        }
        lf2 hA[4];
        float hC[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            hA[i] = lf2{0.f, 0.f}; hC[i] = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const float w = kGauss[k];
                hA[i] = lfma2(w, v[i + k], hA[i]); hC[i] = fmaf(w, d[i + k], hC[i]);
            }
        }
        float4* dA = reinterpret_cast<float4*>(&s_hA[r * kStrH2 + 8 * g]);
        dA[0] = make_float4(hA[0].x, hA[0].y, hA[1].x, hA[1].y); dA[1] = make_float4(hA[2].x, hA[2].y, hA[3].x, hA[3].y);
        *reinterpret_cast<float4*>(&s_hC[r * kStrH1 + 4 * g]) = make_float4(hC[0], hC[1], hC[2], hC[3]);
    }
    __syncthreads();
    const int cc = tid & (kTW - 1), r0 = (tid / kTW) * kRP;
    const int gx = lt.bx * kTW + cc;
    if (gx >= W) return;
    lf2 cA[kRP];
    float c2[kRP];
    {
        lf2 a[kRP + 10];
        float d[kRP + 10];
#pragma unroll
        for (int k = 0; k < kRP + 10; k++) {
            a[k] = *reinterpret_cast<const lf2*>(&s_hA[(r0 + k) * kStrH2 + 2 * cc]);
            d[k] = s_hC[(r0 + k) * kStrH1 + cc];
        }
#pragma unroll
        for (int i = 0; i < kRP; i++) {
            cA[i] = lf2{0.f, 0.f}; c2[i] = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const float w = kGauss[k];
                cA[i] = lfma2(w, a[i + k], cA[i]); c2[i] = fmaf(w, d[i + k], c2[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < kRP; i++) {
        const int gy = lt.by * kTH + r0 + i;
        if (gy >= H) break;
        const size_t pid = (size_t)c * P + (size_t)gy * W + gx;
        const float r = raw[pid], yv = gt[pid];
        const float xv = do_clamp ? clamp01(r) : r;
        const float dssim = cA[i].x + 2.f * xv * cA[i].y + yv * c2[i];   // d(sum of SSIM map)/dx
        const float diff = xv - yv;
        const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
        float g = ((1.f - lambda) * sgn - lambda * dssim) * inv_count;
        if (do_clamp && (r < 0.f || r > 1.f)) g = 0.f;               // clamp backward (inclusive pass-through like torch)
        d_raw[pid] = g * (gscale ? gscale[0] : 1.f);
    }
}

}  // namespace gsr

using namespace gsr;

extern "C" {

size_t gsr_loss_workspace_bytes(int32_t C, int32_t H, int32_t W)
{
    const size_t nb = (size_t)((W + kLB - 1) / kLB) * ((H + kLB - 1) / kLB) * C;
    const size_t maps = (size_t)3 * C * H * W * sizeof(float);
    return ((maps + 255) & ~(size_t)255) + ((nb * 2 * sizeof(float) + 255) & ~(size_t)255);
}
size_t gsr_loss_workspace_bytes_batched(int32_t images, int32_t C, int32_t H, int32_t W) { return gsr_loss_workspace_bytes(images * C, H, W); }

static int loss_forward_impl(const float* render, const float* target, int32_t images, int32_t C, int32_t H, int32_t W, float lambda_dssim,
                             int32_t clamp01_render, void* workspace, float* out3, void* stream, int nout, float* loss_dup = nullptr);
int gsr_loss_forward_batched(const float* render, const float* target, int32_t images, int32_t C, int32_t H, int32_t W, float lambda_dssim,
                             int32_t clamp01_render, void* workspace, float* out3, void* stream)
{
    return loss_forward_impl(render, target, images, C, H, W, lambda_dssim, clamp01_render, workspace, out3, stream, 3);
}
int gsr_loss_forward_terms(const float* render, const float* target, int32_t images, int32_t C, int32_t H, int32_t W, float lambda_dssim,
                           int32_t clamp01_render, void* workspace, float* out6, float* loss_copy, void* stream)
{
    return loss_forward_impl(render, target, images, C, H, W, lambda_dssim, clamp01_render, workspace, out6, stream, 6, loss_copy);
}
static int loss_forward_impl(const float* render, const float* target, int32_t images, int32_t C, int32_t H, int32_t W, float lambda_dssim,
                             int32_t clamp01_render, void* workspace, float* out3, void* stream, int nout, float* loss_dup)
{
    if (!render || !target || !workspace || !out3 || images <= 0 || C <= 0 || H <= 0 || W <= 0) return GSR_ERR_ARG;
    const int CT = images * C;   // channels are independent of one another in both terms: the stack is CT channel planes
    const dim3 grid((W + kTW - 1) / kTW, (H + kTH - 1) / kTH, CT), block(256);
    const size_t maps_bytes = (((size_t)3 * CT * H * W * sizeof(float)) + 255) & ~(size_t)255;
    float* maps = static_cast<float*>(workspace);
    float* partial = reinterpret_cast<float*>(static_cast<uint8_t*>(workspace) + maps_bytes);
    const int nb = (int)(grid.x * grid.y * grid.z);
    hipLaunchKernelGGL(k_loss_fwd, grid, block, 0, (hipStream_t)stream, render, target, H, W, clamp01_render, maps, partial);
    hipLaunchKernelGGL(k_loss_finish, dim3(1), dim3(1024), 0, (hipStream_t)stream, partial, nb, 1.0f / ((float)C * H * W), lambda_dssim, out3, images, nout, loss_dup);
    return hipGetLastError() == hipSuccess ? GSR_OK : GSR_ERR_HIP;
}

int gsr_loss_backward_batched(const float* render, const float* target, int32_t images, int32_t C, int32_t H, int32_t W, float lambda_dssim,
                              int32_t clamp01_render, const void* workspace, const float* grad_loss, float* d_render, void* stream)
{
    if (!render || !target || !workspace || !d_render || images <= 0 || C <= 0 || H <= 0 || W <= 0) return GSR_ERR_ARG;
    const dim3 grid((W + kTW - 1) / kTW, (H + kTH - 1) / kTH, images * C), block(256);
    // every image with ITS OWN normalisation 1 / (C H W): the gradient of the sum of the images' losses
    hipLaunchKernelGGL(k_loss_bwd, grid, block, 0, (hipStream_t)stream, render, target, H, W, clamp01_render,
                       static_cast<const float*>(workspace), grad_loss, 1.0f / ((float)C * H * W), lambda_dssim, d_render);
    return hipGetLastError() == hipSuccess ? GSR_OK : GSR_ERR_HIP;
}

int gsr_loss_forward(const float* render, const float* target, int32_t C, int32_t H, int32_t W, float lambda_dssim,
                     int32_t clamp01_render, void* workspace, float* out3, void* stream)
{
    return gsr_loss_forward_batched(render, target, 1, C, H, W, lambda_dssim, clamp01_render, workspace, out3, stream);
}

int gsr_loss_backward(const float* render, const float* target, int32_t C, int32_t H, int32_t W, float lambda_dssim,
                      int32_t clamp01_render, const void* workspace, const float* grad_loss, float* d_render, void* stream)
{
    return gsr_loss_backward_batched(render, target, 1, C, H, W, lambda_dssim, clamp01_render, workspace, grad_loss, d_render, stream);
}

}  // extern "C"
