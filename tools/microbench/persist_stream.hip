// Feasibility probe (round 5): a PERSISTENT streaming kernel with a bounded footprint per CU -- does it leave room for the
// forward's latency-bound kernels (depth sort: one 1 024-thread / 135 kB workgroup per CU; direct binning) on another stream?
// Three 16-byte streams in, three out (the shape of an Adam update over the f_rest group).  Built by tools/overlap_persist.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

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

template <int U>
__global__ __launch_bounds__(256) void k_persist_stream(const float4* __restrict__ a, const float4* __restrict__ b,
                                                        const float4* __restrict__ c, float4* __restrict__ oa,
                                                        float4* __restrict__ ob, float4* __restrict__ oc, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t q0 = (size_t)blockIdx.x * 256 * U + threadIdx.x; q0 < n4; q0 += stride) {
        float4 x[U], y[U], z[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t q = q0 + (size_t)u * 256;
            if (q < n4) {
                x[u] = a[q];
                y[u] = nt_load4(b + q);
                z[u] = nt_load4(c + q);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t q = q0 + (size_t)u * 256;
            if (q < n4) {
                float4 p = x[u], m = y[u], v = z[u];
                m.x = 0.9f * m.x + 0.1f * p.x; m.y = 0.9f * m.y + 0.1f * p.y; m.z = 0.9f * m.z + 0.1f * p.z; m.w = 0.9f * m.w + 0.1f * p.w;
                v.x = 0.999f * v.x + 0.001f * p.x * p.x; v.y = 0.999f * v.y + 0.001f * p.y * p.y;
                v.z = 0.999f * v.z + 0.001f * p.z * p.z; v.w = 0.999f * v.w + 0.001f * p.w * p.w;
                p.x -= 1e-3f * m.x; p.y -= 1e-3f * m.y; p.z -= 1e-3f * m.z; p.w -= 1e-3f * m.w;
                nt_store4(oa + q, p);
                nt_store4(ob + q, m);
                nt_store4(oc + q, v);
            }
        }
    }
}

extern "C" int persist_stream(const void* a, const void* b, const void* c, void* oa, void* ob, void* oc, size_t n4, int blocks,
                              int unroll, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
#define GO(U) hipLaunchKernelGGL(k_persist_stream<U>, dim3(blocks), dim3(256), 0, s, (const float4*)a, (const float4*)b, \
                                 (const float4*)c, (float4*)oa, (float4*)ob, (float4*)oc, n4)
    switch (unroll) {
        case 1: GO(1); break;
        case 2: GO(2); break;
        case 4: GO(4); break;
        case 8: GO(8); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}
