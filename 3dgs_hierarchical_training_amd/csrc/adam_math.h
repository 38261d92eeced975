// adam_math.h -- the Adam update shared by k_adam (optim_kernels.hip) and the optimizer-in-backward mode of
// k_preprocess_bwd (gsr_kernels.hip).  Update rule = torch.optim.Adam without amsgrad / weight decay, as the reference
// constructs it (/root/reference/scene/gaussian_model_ht.py:275-289):
//   m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#pragma once
#include <hip/hip_runtime.h>

namespace gsr {

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float b1, float b2, float eps, float step_size,
                                         float inv_bc2s)
{
    m = fmaf(g - m, 1.f - b1, m);
    v = fmaf(v, b2, (1.f - b2) * g * g);
    const float denom = sqrtf(v) * inv_bc2s + eps;
    p -= step_size * (m / denom);
}

// Streaming accesses: parameters and moments are touched once per step (1.4 kB per Gaussian), so they carry the `nt` bit and
// do not displace the splats, lists and checkpoints the neighbouring kernels keep in L2 / MALL.
typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float4* a)
{
    const nt_f4 r = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(a));
    return make_float4(r.x, r.y, r.z, r.w);
}
__device__ __forceinline__ void nt_store4(float4* a, const float4& x)
{
    nt_f4 r = {x.x, x.y, x.z, x.w};
    __builtin_nontemporal_store(r, reinterpret_cast<nt_f4*>(a));
}

}  // namespace gsr
