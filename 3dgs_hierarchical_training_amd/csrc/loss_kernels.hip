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

constexpr int kLB = 16;           // output tile
constexpr int kHalo = 5;          // window 11
constexpr int kLIn = kLB + 2 * kHalo;   // 26
constexpr float kC1 = 0.01f * 0.01f, kC2 = 0.03f * 0.03f;

// exp(-(i-5)^2 / (2*1.5^2)) normalised, i = 0..10 (losses.py:147-150)
__device__ __constant__ float kGauss[11] = {0.0010283801f, 0.0075987581f, 0.0360007721f, 0.1093606895f, 0.2130055377f,
                                            0.2660117249f, 0.2130055377f, 0.1093606895f, 0.0360007721f, 0.0075987581f,
                                            0.0010283801f};

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

// Two moments per register pair: (x, y) and (x^2, y^2) ride through the separable window as float2, so a tap is one
// 8-byte LDS read and v_pk_fma_f32 instead of two reads and two fmas (per-component fma: same rounding as scalar code).
typedef float lf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ lf2 lfma2(float w, lf2 v, lf2 acc) { return __builtin_elementwise_fma(lf2{w, w}, v, acc); }

// grid (ceil(W/16), ceil(H/16), C); block 16x16.  maps: [3][C][H][W] = dS/dmu_x, dS/dE[x^2], dS/dE[xy]
__global__ __launch_bounds__(256) void k_loss_fwd(const float* __restrict__ raw, const float* __restrict__ gt, int H, int W,
                                                  int do_clamp, float* __restrict__ maps, float* __restrict__ partial)
{
    __shared__ lf2 sxy[kLIn][kLIn + 1];                       // (x, y) interleaved
    __shared__ lf2 shA[kLIn][kLB + 1], shB[kLIn][kLB + 1];     // row-filtered (x, y) and (x^2, y^2)
    __shared__ float shC[kLIn][kLB + 1];                       // row-filtered x y
    __shared__ float s_red[2][4];
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * kLB + tx;
    const int c = blockIdx.z;
    const size_t P = (size_t)H * W;
    const float* x = raw + (size_t)c * P;
    const float* y = gt + (size_t)c * P;
    const int ox = blockIdx.x * kLB - kHalo, oy = blockIdx.y * kLB - kHalo;
    for (int i = tid; i < kLIn * kLIn; i += 256) {
        const int iy = i / kLIn, ix = i - iy * kLIn;
        const int gy = oy + iy, gx = ox + ix;
        float xv = 0.f, yv = 0.f;
        if (gx >= 0 && gx < W && gy >= 0 && gy < H) {
            xv = x[(size_t)gy * W + gx];
            if (do_clamp) xv = clamp01(xv);
            yv = y[(size_t)gy * W + gx];
        }
        sxy[iy][ix] = lf2{xv, yv};
    }
    __syncthreads();
    for (int i = tid; i < kLIn * kLB; i += 256) {
        const int r = i / kLB, cc = i - r * kLB;
        lf2 hA = {0.f, 0.f}, hB = {0.f, 0.f};
        float hC = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = kGauss[k];
            const lf2 v = sxy[r][cc + k];
            hA = lfma2(w, v, hA); hB = lfma2(w, v * v, hB); hC = fmaf(w, v.x * v.y, hC);
        }
        shA[r][cc] = hA; shB[r][cc] = hB; shC[r][cc] = hC;
    }
    __syncthreads();
    lf2 mA = {0.f, 0.f}, mB = {0.f, 0.f};
    float e12 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const float w = kGauss[k];
        mA = lfma2(w, shA[ty + k][tx], mA); mB = lfma2(w, shB[ty + k][tx], mB);
        e12 = fmaf(w, shC[ty + k][tx], e12);
    }
    const float mu1 = mA.x, mu2 = mA.y, e11 = mB.x, e22 = mB.y;
    const int gx = blockIdx.x * kLB + tx, gy = blockIdx.y * kLB + ty;
    float ssim = 0.f, l1 = 0.f;
    if (gx < W && gy < H) {
        const float s11 = e11 - mu1 * mu1, s22 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
        const float A = 2.f * mu1 * mu2 + kC1, B = 2.f * s12 + kC2;
        const float Cc = mu1 * mu1 + mu2 * mu2 + kC1, D = s11 + s22 + kC2;
        const float iCD = 1.f / (Cc * D);
        ssim = A * B * iCD;
        // partial derivatives of S w.r.t. the three moments that depend on x (mu_x, E[x^2], E[xy])
        const float dS_dmu = 2.f * mu2 * (B - A) * iCD - 2.f * mu1 * ssim / Cc + 2.f * mu1 * ssim / D;
        const float dS_de11 = -ssim / D;
        const float dS_de12 = 2.f * A * iCD;
        const size_t CP = (size_t)gridDim.z * P, pid = (size_t)c * P + (size_t)gy * W + gx;
        maps[pid] = dS_dmu; maps[CP + pid] = dS_de11; maps[2 * CP + pid] = dS_de12;
        const lf2 ctr = sxy[ty + kHalo][tx + kHalo];
        l1 = fabsf(ctr.x - ctr.y);
    }
    // block reduction (wave shuffles, then 4 partials)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { ssim += __shfl_xor(ssim, off, 64); l1 += __shfl_xor(l1, off, 64); }
    if ((tid & 63) == 0) { s_red[0][tid >> 6] = ssim; s_red[1][tid >> 6] = l1; }
    __syncthreads();
    if (tid == 0) {
        const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[2 * b] = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
        partial[2 * b + 1] = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
    }
}

// single block: deterministic reduction of the per-block partials -> out[0] = loss, out[1] = mean ssim, out[2] = mean l1
__global__ __launch_bounds__(1024) void k_loss_finish(const float* __restrict__ partial, int nblocks, float inv_count, float lambda,
                                                      float* __restrict__ out)
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
        out[0] = (float)((1.0 - lambda) * ml + lambda * (1.0 - ms));
        out[1] = (float)ms; out[2] = (float)ml;
    }
}

__global__ __launch_bounds__(256) void k_loss_bwd(const float* __restrict__ raw, const float* __restrict__ gt, int H, int W,
                                                  int do_clamp, const float* __restrict__ maps, const float* __restrict__ gscale,
                                                  float inv_count, float lambda, float* __restrict__ d_raw)
{
    __shared__ lf2 smA[kLIn][kLIn + 1];     // (dS/dmu_x, dS/dE[x^2])
    __shared__ float smC[kLIn][kLIn + 1];   // dS/dE[xy]
    __shared__ lf2 shA[kLIn][kLB + 1];
    __shared__ float shC[kLIn][kLB + 1];
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * kLB + tx;
    const int c = blockIdx.z;
    const size_t P = (size_t)H * W, CP = (size_t)gridDim.z * P;
    const int ox = blockIdx.x * kLB - kHalo, oy = blockIdx.y * kLB - kHalo;
    for (int i = tid; i < kLIn * kLIn; i += 256) {
        const int iy = i / kLIn, ix = i - iy * kLIn;
        const int gy = oy + iy, gx = ox + ix;
        float a = 0.f, b = 0.f, d = 0.f;
        if (gx >= 0 && gx < W && gy >= 0 && gy < H) {
            const size_t pid = (size_t)c * P + (size_t)gy * W + gx;
            a = maps[pid]; b = maps[CP + pid]; d = maps[2 * CP + pid];
        }
        smA[iy][ix] = lf2{a, b}; smC[iy][ix] = d;
    }
    __syncthreads();
    for (int i = tid; i < kLIn * kLB; i += 256) {
        const int r = i / kLB, cc = i - r * kLB;
        lf2 hA = {0.f, 0.f};
        float hC = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = kGauss[k];
            hA = lfma2(w, smA[r][cc + k], hA); hC = fmaf(w, smC[r][cc + k], hC);
        }
        shA[r][cc] = hA; shC[r][cc] = hC;
    }
    __syncthreads();
    const int gx = blockIdx.x * kLB + tx, gy = blockIdx.y * kLB + ty;
    if (gx >= W || gy >= H) return;
    lf2 cA = {0.f, 0.f};
    float c2 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const float w = kGauss[k];
        cA = lfma2(w, shA[ty + k][tx], cA); c2 = fmaf(w, shC[ty + k][tx], c2);
    }
    const float c0 = cA.x, c1 = cA.y;
    const size_t pid = (size_t)c * P + (size_t)gy * W + gx;
    const float r = raw[pid], yv = gt[pid];
    const float xv = do_clamp ? clamp01(r) : r;
    const float dssim = c0 + 2.f * xv * c1 + yv * c2;           // d(sum of SSIM map)/dx
    const float diff = xv - yv;
    const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
    float g = ((1.f - lambda) * sgn - lambda * dssim) * inv_count;
    if (do_clamp && (r < 0.f || r > 1.f)) g = 0.f;               // clamp backward (inclusive pass-through like torch)
    d_raw[pid] = g * (gscale ? gscale[0] : 1.f);
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

int gsr_loss_forward(const float* render, const float* target, int32_t C, int32_t H, int32_t W, float lambda_dssim,
                     int32_t clamp01_render, void* workspace, float* out3, void* stream)
{
    if (!render || !target || !workspace || !out3 || C <= 0 || H <= 0 || W <= 0) return GSR_ERR_ARG;
    const dim3 grid((W + kLB - 1) / kLB, (H + kLB - 1) / kLB, C), block(kLB, kLB);
    const size_t maps_bytes = (((size_t)3 * C * H * W * sizeof(float)) + 255) & ~(size_t)255;
    float* maps = static_cast<float*>(workspace);
    float* partial = reinterpret_cast<float*>(static_cast<uint8_t*>(workspace) + maps_bytes);
    const int nb = (int)(grid.x * grid.y * grid.z);
    hipLaunchKernelGGL(k_loss_fwd, grid, block, 0, (hipStream_t)stream, render, target, H, W, clamp01_render, maps, partial);
    hipLaunchKernelGGL(k_loss_finish, dim3(1), dim3(1024), 0, (hipStream_t)stream, partial, nb, 1.0f / ((float)C * H * W), lambda_dssim, out3);
    return hipGetLastError() == hipSuccess ? GSR_OK : GSR_ERR_HIP;
}

int gsr_loss_backward(const float* render, const float* target, int32_t C, int32_t H, int32_t W, float lambda_dssim,
                      int32_t clamp01_render, const void* workspace, const float* grad_loss, float* d_render, void* stream)
{
    if (!render || !target || !workspace || !d_render || C <= 0 || H <= 0 || W <= 0) return GSR_ERR_ARG;
    const dim3 grid((W + kLB - 1) / kLB, (H + kLB - 1) / kLB, C), block(kLB, kLB);
    hipLaunchKernelGGL(k_loss_bwd, grid, block, 0, (hipStream_t)stream, render, target, H, W, clamp01_render,
                       static_cast<const float*>(workspace), grad_loss, 1.0f / ((float)C * H * W), lambda_dssim, d_render);
    return hipGetLastError() == hipSuccess ? GSR_OK : GSR_ERR_HIP;
}

}  // extern "C"
