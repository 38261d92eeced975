// optim_kernels.hip -- multi-tensor Adam step in ONE launch ("next" row f-2 of SURVEY.md section 8).
//
// The reference updates its six parameter groups with torch.optim.Adam(l, lr=0.0, eps=1e-15)
// (/root/reference/scene/gaussian_model_ht.py:275-289; per-group learning rates
// /root/reference/arguments/__init__.py:116-131).  That is 59 floats per Gaussian; the update is a pure HBM
// stream: read p, g, m, v and write p, m, v = 28 B per float = 1652 B per Gaussian.  This kernel walks all groups
// in a single grid with 16-byte accesses.  Update rule = torch's (non-amsgrad, no weight decay):
//   m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gsr.h"
#include "adam_math.h"

namespace gsr {

constexpr int kAdamThreads = 256;
constexpr int kAdamVec = 4;                 // floats per thread per step (16-byte accesses)
constexpr int kAdamChunk = kAdamThreads * kAdamVec * 4;   // floats per block

struct AdamBatch {
    GsrAdamTensor t[GSR_ADAM_MAX_TENSORS];
    uint32_t block_start[GSR_ADAM_MAX_TENSORS + 1];
    int count;
    float beta1, beta2, eps, bc1, bc2_sqrt;
};

__global__ __launch_bounds__(kAdamThreads) void k_adam(AdamBatch B)
{
    int ti = 0;
#pragma unroll
    for (int k = 1; k < GSR_ADAM_MAX_TENSORS; k++)
        if (k < B.count && blockIdx.x >= B.block_start[k]) ti = k;
    const GsrAdamTensor t = B.t[ti];
    const uint64_t base = (uint64_t)(blockIdx.x - B.block_start[ti]) * kAdamChunk;
    const float step_size = t.lr / B.bc1, inv_bc2s = 1.f / B.bc2_sqrt;
    const bool aligned = ((((uintptr_t)t.param | (uintptr_t)t.grad | (uintptr_t)t.exp_avg | (uintptr_t)t.exp_avg_sq) & 15) == 0);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint64_t i = base + (uint64_t)(r * kAdamThreads + threadIdx.x) * kAdamVec;
        if (i >= t.n) break;
        if (aligned && i + kAdamVec <= t.n) {
            float4 p = nt_load4(reinterpret_cast<const float4*>(t.param + i));
            const float4 g = nt_load4(reinterpret_cast<const float4*>(t.grad + i));
            float4 m = nt_load4(reinterpret_cast<const float4*>(t.exp_avg + i));
            float4 v = nt_load4(reinterpret_cast<const float4*>(t.exp_avg_sq + i));
            adam_one(p.x, g.x, m.x, v.x, B.beta1, B.beta2, B.eps, step_size, inv_bc2s);
            adam_one(p.y, g.y, m.y, v.y, B.beta1, B.beta2, B.eps, step_size, inv_bc2s);
            adam_one(p.z, g.z, m.z, v.z, B.beta1, B.beta2, B.eps, step_size, inv_bc2s);
            adam_one(p.w, g.w, m.w, v.w, B.beta1, B.beta2, B.eps, step_size, inv_bc2s);
            nt_store4(reinterpret_cast<float4*>(t.param + i), p);
            nt_store4(reinterpret_cast<float4*>(t.exp_avg + i), m);
            nt_store4(reinterpret_cast<float4*>(t.exp_avg_sq + i), v);
        } else {
            for (uint64_t k = i; k < t.n && k < i + kAdamVec; k++) {
                float p = t.param[k], m = t.exp_avg[k], v = t.exp_avg_sq[k];
                adam_one(p, t.grad[k], m, v, B.beta1, B.beta2, B.eps, step_size, inv_bc2s);
                t.param[k] = p; t.exp_avg[k] = m; t.exp_avg_sq[k] = v;
            }
        }
    }
}

}  // namespace gsr

using namespace gsr;

extern "C" int gsr_adam_step(const GsrAdamTensor* tensors, int32_t count, float beta1, float beta2, float eps, int64_t step,
                             void* stream)
{
    if (!tensors || count <= 0 || count > GSR_ADAM_MAX_TENSORS || step <= 0) return GSR_ERR_ARG;
    AdamBatch B;
    uint32_t blocks = 0;
    for (int k = 0; k < count; k++) {
        B.t[k] = tensors[k];
        if (tensors[k].n && (!tensors[k].param || !tensors[k].grad || !tensors[k].exp_avg || !tensors[k].exp_avg_sq)) return GSR_ERR_ARG;
        B.block_start[k] = blocks;
        blocks += (uint32_t)((tensors[k].n + kAdamChunk - 1) / kAdamChunk);
    }
    for (int k = count; k <= GSR_ADAM_MAX_TENSORS; k++) B.block_start[k] = blocks;
    B.count = count; B.beta1 = beta1; B.beta2 = beta2; B.eps = eps;
    B.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    B.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    if (blocks == 0) return GSR_OK;
    hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(kAdamThreads), 0, (hipStream_t)stream, B);
    return hipGetLastError() == hipSuccess ? GSR_OK : GSR_ERR_HIP;
}
