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

// grid (ceil(W/16), ceil(H/16), C); block 16x16.  maps: [3][C][H][W] = dS/dmu_x, dS/dE[x^2], dS/dE[xy]
__global__ __launch_bounds__(256) void k_loss_fwd(const float* __restrict__ raw, const float* __restrict__ gt, int H, int W,
                                                  int do_clamp, float* __restrict__ maps, float* __restrict__ partial)
{
    __shared__ float sx[kLIn][kLIn + 1], sy[kLIn][kLIn + 1];
    __shared__ float sh[5][kLIn][kLB + 1];
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
        sx[iy][ix] = xv; sy[iy][ix] = yv;
    }
    __syncthreads();
    for (int i = tid; i < kLIn * kLB; i += 256) {
        const int r = i / kLB, cc = i - r * kLB;
        float h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f, h4 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = kGauss[k], xv = sx[r][cc + k], yv = sy[r][cc + k];
            h0 = fmaf(w, xv, h0); h1 = fmaf(w, yv, h1);
            h2 = fmaf(w, xv * xv, h2); h3 = fmaf(w, yv * yv, h3); h4 = fmaf(w, xv * yv, h4);
        }
        sh[0][r][cc] = h0; sh[1][r][cc] = h1; sh[2][r][cc] = h2; sh[3][r][cc] = h3; sh[4][r][cc] = h4;
    }
    __syncthreads();
    float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const float w = kGauss[k];
        mu1 = fmaf(w, sh[0][ty + k][tx], mu1); mu2 = fmaf(w, sh[1][ty + k][tx], mu2);
        e11 = fmaf(w, sh[2][ty + k][tx], e11); e22 = fmaf(w, sh[3][ty + k][tx], e22);
        e12 = fmaf(w, sh[4][ty + k][tx], e12);
    }
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
        l1 = fabsf(sx[ty + kHalo][tx + kHalo] - sy[ty + kHalo][tx + kHalo]);
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
    __shared__ float sm[3][kLIn][kLIn + 1];
    __shared__ float sh[3][kLIn][kLB + 1];
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
        sm[0][iy][ix] = a; sm[1][iy][ix] = b; sm[2][iy][ix] = d;
    }
    __syncthreads();
    for (int i = tid; i < kLIn * kLB; i += 256) {
        const int r = i / kLB, cc = i - r * kLB;
        float h0 = 0.f, h1 = 0.f, h2 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = kGauss[k];
            h0 = fmaf(w, sm[0][r][cc + k], h0); h1 = fmaf(w, sm[1][r][cc + k], h1); h2 = fmaf(w, sm[2][r][cc + k], h2);
        }
        sh[0][r][cc] = h0; sh[1][r][cc] = h1; sh[2][r][cc] = h2;
    }
    __syncthreads();
    const int gx = blockIdx.x * kLB + tx, gy = blockIdx.y * kLB + ty;
    if (gx >= W || gy >= H) return;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const float w = kGauss[k];
        c0 = fmaf(w, sh[0][ty + k][tx], c0); c1 = fmaf(w, sh[1][ty + k][tx], c1); c2 = fmaf(w, sh[2][ty + k][tx], c2);
    }
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
