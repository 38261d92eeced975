// gather_calib.hip -- what FETCH_SIZE / WRITE_SIZE report for THIS library's access patterns.
// The guide calibrates FETCH_SIZE only on wide coalesced streams (x2 on gfx950).  The blend kernels gather 48-byte splat records at
// 48-byte stride with three 16-byte loads per lane, write 4-byte-per-lane planes, and flush gradients with float atomics on 9 of
// a record's 12 words from 16 lanes per record.  Each pattern below moves a KNOWN number of bytes over a table far larger than
// L2 + Infinity Cache's useful share; run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (tools/microbench/run_gather_calib.sh)
// and compare.  One kernel per pattern, kernel names carry the pattern; the program prints the true byte counts as JSON.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>
#include <numeric>
#include <random>
#include <vector>

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Rec { float4 a, b, c; };   // 48 bytes, 16-byte aligned: the splat record's layout

// reference pattern: every lane reads 16 consecutive bytes (the guide's calibrated case)
__global__ void k_stream_read16(const float4* __restrict__ src, size_t n4, float* __restrict__ sink)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) sink[0] = acc;
}

// 4 bytes per lane, consecutive (the list reads of the blend)
__global__ void k_stream_read4(const float* __restrict__ src, size_t n, float* __restrict__ sink)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
    if (acc == 123.456f) sink[0] = acc;
}

// the blend's staging: lane j gathers record idx[j] with three 16-byte loads
__global__ void k_gather48(const Rec* __restrict__ recs, const uint32_t* __restrict__ idx, size_t n, float* __restrict__ sink)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4* p = reinterpret_cast<const float4*>(recs + idx[i]);
        const float4 a = p[0], b = p[1], c = p[2];
        acc += a.x + b.y + c.z;
    }
    if (acc == 123.456f) sink[0] = acc;
}

// 4-byte-per-lane plane writes (image state, outputs, checkpoints)
__global__ void k_stream_write4(float* __restrict__ dst, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (float)i;
}

// 16-byte-per-lane writes (the per-Gaussian kernels' streams)
__global__ void k_stream_write16(float4* __restrict__ dst, size_t n4)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}

// the backward blend's flush: 16 lanes per record, lanes 0..8 add to the record's first nine words
__global__ void k_atomic_flush(float* __restrict__ recs12, const uint32_t* __restrict__ idx, size_t n)
{
    const int r = threadIdx.x & 15;
    for (size_t g = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 4; g < n; g += ((size_t)gridDim.x * blockDim.x) >> 4)
        if (r < 9) atomicAdd(recs12 + (size_t)idx[g] * 12 + r, 1.0f);
}

int main(int argc, char** argv)
{
    const size_t NREC = (size_t)48 << 20;          // 48 Mi records x 48 B = 2.25 GiB table (L2 32 MiB, Infinity Cache 256 MiB)
    const size_t NG = (size_t)8 << 20;             // gathered / flushed records per launch
    const size_t NSTREAM = (size_t)1 << 30;        // bytes of the streaming patterns
    Rec* recs; uint32_t* idx; float *stream, *sink;
    HIPCHECK(hipMalloc(&recs, NREC * sizeof(Rec)));
    HIPCHECK(hipMalloc(&idx, NG * 4));
    HIPCHECK(hipMalloc(&stream, NSTREAM));
    HIPCHECK(hipMalloc(&sink, 256));
    HIPCHECK(hipMemset(recs, 0, NREC * sizeof(Rec)));
    HIPCHECK(hipMemset(stream, 0, NSTREAM));
    std::vector<uint32_t> h(NG);
    std::mt19937_64 rng(12345);
    // distinct random records: each is touched once per launch (what a frame's ~1 M staged instances of 1 M Gaussians are, scaled up)
    {
        std::vector<uint32_t> all(NREC);
        std::iota(all.begin(), all.end(), 0u);
        for (size_t i = 0; i < NG; i++) { const size_t j = i + rng() % (NREC - i); std::swap(all[i], all[j]); h[i] = all[i]; }
    }
    HIPCHECK(hipMemcpy(idx, h.data(), NG * 4, hipMemcpyHostToDevice));
    const dim3 grid(4096), blk(256);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_stream_read16, grid, blk, 0, 0, reinterpret_cast<const float4*>(stream), NSTREAM / 16, sink);
        hipLaunchKernelGGL(k_stream_read4, grid, blk, 0, 0, stream, NSTREAM / 4, sink);
        hipLaunchKernelGGL(k_gather48, grid, blk, 0, 0, recs, idx, NG, sink);
        hipLaunchKernelGGL(k_stream_write4, grid, blk, 0, 0, stream, NSTREAM / 4);
        hipLaunchKernelGGL(k_stream_write16, grid, blk, 0, 0, reinterpret_cast<float4*>(stream), NSTREAM / 16);
        hipLaunchKernelGGL(k_atomic_flush, grid, blk, 0, 0, reinterpret_cast<float*>(recs), idx, NG);
        HIPCHECK(hipDeviceSynchronize());
    }
    // expected bytes per launch.  A 48-byte record at 48-byte stride lies in one 64-byte sector half of the time and in two the other
    // half (96 B per record on average); at 128-byte granularity 1.25 lines = 160 B.
    printf("{\"k_stream_read16\": {\"read\": %zu}, \"k_stream_read4\": {\"read\": %zu}, "
           "\"k_gather48\": {\"read_requested\": %zu, \"read_idx\": %zu, \"read_at_64B_sectors\": %zu, \"read_at_128B_lines\": %zu}, "
           "\"k_stream_write4\": {\"write\": %zu}, \"k_stream_write16\": {\"write\": %zu}, "
           "\"k_atomic_flush\": {\"words_updated_bytes\": %zu, \"read_idx\": %zu, \"rmw_at_64B_sectors\": %zu, \"records\": %zu}}\n",
           NSTREAM, NSTREAM, NG * 48, NG * 4, NG * 96, NG * 160, NSTREAM, NSTREAM, NG * 36, (NG * 4), NG * 64, NG);
    return 0;
}
