// gsr_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) + the C ABI of include/gsr.h.
//
// Pipeline (DESIGN.md section 4 has a cell per kernel with its time and roof; three binning routes build the same lists bit for bit):
//   forward   K1 k_preprocess      per Gaussian: cull, project, cov2D, conic, radius, SH colour -> 48-B Splat, 8-B TileRec, depth key
//                                  (in the fused training loop its work rides in K9: "prepare in backward")
//             K2 onesweep sort     depth keys over N: histogram + three 9-bit passes (radix_sort.h)        -> depth order
//             K3-K6 direct binning k_chunk_counts / k_chunk_scan1 / k_chunk_scan2 / k_chunk_scatter: every (Gaussian, tile) pair written
//                                  once, at its final place (frames of up to 4 096 tiles)
//             K3s-K6s emit path    k_tile_counts / k_block_scan, k_emit, onesweep sort of 16-bit tile keys over R, k_tile_ranges
//                                  (larger frames, batched renders)
//             K2t tile-sort route  no K2: binning in INDEX order, then k_tile_sort_wave / k_tile_sort<1024> order every tile's segment by
//                                  depth key, stable (models with short tile lists)
//             K7 k_blend_fwd_w6    one wave per 8x8 pixel block, front-to-back compositing, LDS-staged lists, checkpoints
//   backward  K8p k_bwd_prologue   zero-fill of the per-Gaussian accumulators + the backward blend's work items
//             K8 k_blend_bwd2      persistent workgroups, front-to-back replay from checkpoints, DPP reductions, one atomic set per (tile, Gaussian)
//             K9 k_preprocess_bwd  per Gaussian: conic / cov2D / cov3D / projection / SH chain rule (float64), optionally the Adam update of
//                                  all six parameter groups and the next render's K1
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <type_traits>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "../../include/gsr.h"
#include "gsr_math.h"
#include "adam_math.h"
#include "radix_sort.h"
#include "blend_common.h"

#ifndef GSR_K9_UNROLL
#define GSR_K9_UNROLL 2
#endif
#ifndef GSR_K9_PREFETCH
#define GSR_K9_PREFETCH 2      // stream elements (16 B of each moment) per thread that k_preprocess_bwd requests in front of its derivative chain.
                               // Measured at 1 M (same-box A/B, two boxes): 0 -> 2: 320 / 331 -> 308 / 315 us and 332 / 321 -> 328 / 318 us; 3, 4, 6: 362 - 381 us --
                               // the PREP kernel stands at 167 VGPRs with 2 (three waves per SIMD, what its LDS tile allows anyway) and drops to two waves beyond
#endif
namespace gsr {

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int g_blend_ppt = 0;   // 0 = default
static int g_bwd_ppt = 0;
// 1 (default) = onesweep (decoupled look-back) for the depth sort over N, histogram+scan+scatter for the tile
// sort over R (measured: 115 vs 140 us and 164 vs 157 us); 0 = three-kernel passes everywhere; 2 = onesweep everywhere
// prepare in backward: up to this many Gaussians the per-Gaussian backward kernel also counts the digits of the next depth sort
// (saves that sort's histogram launch: ~7 us + a launch gap on small models); above, its ~2 global adds per key cost more inside
// the HBM-bound kernel than the histogram kernel does on its own (1 M: +20 us against -8 us).  Both sides of the hand-over
// evaluate the same predicate on N.
static int g_prep_hist_max_n = 262144;
static int g_small_sort9 = 1;   // smallest N for which a forward with its own preprocess sorts depths in three 9-bit passes (0 = only above prep_hist_max_n).
                                // Measured (same-box A/B, tools/ab_130k.sh): depth sort 53 -> 45 us at 130 k, 44 -> 37 us at 20 k
static inline bool prep_counts_digits(int N) { return N <= g_prep_hist_max_n; }
static int g_poll_iters = 400000;   // bound of gsr_forward's busy-wait for R in units of ~50 ns (20 ms); 0 = event record + hipEventSynchronize instead
static int g_emit_hist = 1;   // 1: k_emit counts the tile sort's digits (no histogram launch); 0: k_radix_ghist
static int g_sort_algo = 2;   // 2: onesweep for both sorts; 1: onesweep depth sort + hist/scan/scatter tile sort; 0: hist/scan/scatter

static int fail(int code, const char* fmt, const char* detail = "")
{
    snprintf(g_err, sizeof g_err, fmt, detail);
    return code;
}

#define GSR_HIP(expr)                                                         \
    do {                                                                      \
        hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess) return fail(GSR_ERR_HIP, #expr ": %s", hipGetErrorString(e_)); \
    } while (0)

static inline void cpu_relax()
{
#if !defined(__HIP_DEVICE_COMPILE__) && (defined(__x86_64__) || defined(__i386__))
    __builtin_ia32_pause();
#endif
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// Optional per-kernel timing with HIP events on the caller's stream (bench.py roofline leg).
enum ProfId { P_PRE_FWD, P_SORT_DEPTH, P_SCAN, P_EMIT, P_SORT_TILE, P_RANGES, P_BLEND_FWD, P_BLEND_BWD, P_PRE_BWD, P_CUT_REPAIR, P_COUNT };
static const char* kProfNames[P_COUNT] = {"preprocess_fwd", "sort_depth", "scan", "emit", "sort_tile", "ranges",
                                          "blend_fwd", "blend_bwd", "preprocess_bwd", "cut_repair"};
static int g_profile = 0;
static std::atomic<unsigned> g_profile_tick{0};   // mode 3: every third launch of the forward blend is timed
struct ProfPair { hipEvent_t a, b; };
static std::mutex g_prof_mutex;
static std::vector<ProfPair> g_prof_events[P_COUNT];
struct ProfScope {
    int id; hipStream_t st; hipEvent_t a = nullptr, b = nullptr;
    ProfScope(int id_, hipStream_t st_) : id(id_), st(st_)
    {
        if (!g_profile || id >= P_COUNT || (g_profile >= 2 && id != P_BLEND_FWD)) return;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
        (void)hipEventRecord(a, st);
    }
    ~ProfScope()
    {
        if (!a) return;
        (void)hipEventRecord(b, st);
        std::lock_guard<std::mutex> lk(g_prof_mutex);
        g_prof_events[id].push_back({a, b});
    }
};

// Batched rendering of independent models (GsrBatch, include/gsr.h): B models live in ONE parameter store, model b owning the
// 128-Gaussian blocks [first_block[b], first_block[b + 1]), each with its own camera (viewmatrix / projmatrix / campos /
// points_transform are arrays of B entries) and its own W x H image.  The B images are handled as one "tall" image of
// B * tiles_y tile rows: tile id = b * T + local tile, so a tile's list only ever holds Gaussians of its own model and ONE
// global depth sort orders every model's lists at once.  B <= 1 = the ordinary single-model render.
constexpr int kMaxBatch = 16;
struct BatchDev {
    int B;
    int first_block[kMaxBatch + 1];
};

struct CamParams {
    const float *vm, *pm, *campos;
    float tanfovx, tanfovy, scale_mod;
    int W, H, D, M;
    const float* xf;   // optional rigid / affine transform of the means (3x4 row-major), see GsrForwardArgs::points_transform
    BatchDev bt;
};

// Which model a 128-Gaussian block belongs to; moves the camera pointers of `p` (a kernel's own copy) to that model's entries.
// Block-uniform: scalar compares against kernel arguments.
__device__ __forceinline__ int select_view(CamParams& p, int block)
{
    if (p.bt.B <= 1) return 0;
    int b = 0;
#pragma unroll
    for (int q = 1; q < kMaxBatch; q++) b += (q < p.bt.B && block >= p.bt.first_block[q]) ? 1 : 0;
    p.vm += 16 * b; p.pm += 16 * b;
    if (p.campos) p.campos += 3 * b;
    if (p.xf) p.xf += 12 * b;
    return b;
}

// p' = M [p; 1]: the in-kernel form of `P.retr().act(xyz)` (/root/reference/scene/gaussian_model_ht.py:135-148)
__device__ __forceinline__ void apply_points_transform(const float* __restrict__ xf, float m[3])
{
    if (!xf) return;
    const float x = m[0], y = m[1], z = m[2];
#pragma unroll
    for (int r = 0; r < 3; r++) m[r] = fmaf(xf[4 * r], x, fmaf(xf[4 * r + 1], y, fmaf(xf[4 * r + 2], z, xf[4 * r + 3])));
}

__device__ __forceinline__ Camera load_camera(const CamParams& p)
{
    Camera c;
#pragma unroll
    for (int k = 0; k < 16; k++) { c.vm[k] = p.vm[k]; c.pm[k] = p.pm[k]; }
    c.cam[0] = p.campos ? p.campos[0] : 0.f; c.cam[1] = p.campos ? p.campos[1] : 0.f; c.cam[2] = p.campos ? p.campos[2] : 0.f;
    c.tanfovx = p.tanfovx; c.tanfovy = p.tanfovy;
    c.fx = p.W / (2.f * p.tanfovx); c.fy = p.H / (2.f * p.tanfovy);
    c.scale_mod = p.scale_mod; c.W = p.W; c.H = p.H;
    c.tiles_x = (p.W + kTile - 1) / kTile; c.tiles_y = (p.H + kTile - 1) / kTile;
    c.D = p.D; c.M = p.M;
    return c;
}

// ---- transposed ("butterfly") wave reduction of NV values --------------------------------------------------
// Reducing NV values one by one costs 6 DPP adds each.  Instead, at every halving step two VALUES are paired:
// half of the lanes keep value a and receive the partner lane's a, the other half keep b and receive b, so one
// DPP add retires half a step of TWO values.  After four steps a single register holds, per lane, the 16-lane
// row sum of the value its low lane bits select; two cross-row shuffles finish it.  9 values: 27 VALU + 2
// shuffles instead of 54 DPP adds, and the result is written by 9 lanes in ONE LDS store.
//   quad_perm [1,0,3,2] = 0xB1 (lane^1), [2,3,0,1] = 0x4E (lane^2), row_ror:4 = 0x124, row_ror:8 = 0x128
template <int CTRL>
__device__ __forceinline__ float bfly_pair(float a, float b, bool hi)
{
    const float keep = hi ? b : a, send = hi ? a : b;
    return keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ float bfly_single(float a)
{
    return a + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), CTRL, 0xf, 0xf, false));
}
// ---- the reduction with the selects folded into DPP write masks (round 3) ---------------------------------------------------------
// Written with bfly_pair at all four levels (keep = hi ? b : a, send = hi ? a : b: two selects and one DPP add per pair) the tree
// costs 11 DPP adds + 16 selects, and a select with an SGPR-pair mask is a double-pass instruction on this chip (4.2 SIMD cycles,
// like a DPP add; calibrated, profiles/r03_valu_calib*): rounds 1-2 ran that form (blend backward 210 us; 203 with this one).  A DPP instruction can disable its WRITE per
// bank of four lanes, so for the two row_ror levels the pair step needs none:
//     out = a + ror(a)                    (all lanes)
//     out = b + ror(b)   bank_mask = hi   (the lanes that keep b overwrite)
// Those levels therefore take the wide end of the tree -- row_ror:4 first (hi = lane bit 2: banks 1, 3), then row_ror:8 (hi = bit 3:
// banks 2, 3; it must come second, a rotation by 4 carries into bit 3) -- and the quad_perm levels, whose lanes cannot be masked by
// bank, the narrow end: 17 DPP adds + 4 selects instead of 11 + 16.  Returns t; lane l < 16 holds the total of value
// reduce2_slot<NV>(l) (or none: -1).  Inline asm: the masked second write of `out` cannot be expressed through the builtin.
template <int NV>
__device__ __forceinline__ int reduce2_slot(int lane)
{
    if (lane >= 16) return -1;
    if (!(lane & 1)) return ((lane >> 2) & 1) + 2 * ((lane >> 3) & 1) + 4 * ((lane >> 1) & 1);
    if (lane == 1) return 8;
    return (NV == 10 && lane == 5) ? 9 : -1;
}
template <int NV>
__device__ __forceinline__ float wave_reduce_transposed2(const float* v, int lane)
{
    float q03, q47, q8;
#if defined(__HIP_DEVICE_COMPILE__)
    float r01, r23, r45, r67, r8;
    const float v9 = NV == 10 ? v[NV - 1] : 0.f;
    // (s_nop 1 in front: the inputs were just written by VALU instructions and a DPP read of a VGPR needs two wait states; every
    //  DPP read inside the block is at least four instructions behind the write it depends on; s_nop 1 behind: the builtins below
    //  read q03 / q47 / q8 through DPP)
    if (NV == 10)
        asm volatile("s_nop 1\n"
                     "v_add_f32_dpp %3, %8, %8 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %9, %9 row_ror:4 row_mask:0xf bank_mask:0xa\n"
                     "v_add_f32_dpp %4, %10, %10 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %4, %11, %11 row_ror:4 row_mask:0xf bank_mask:0xa\n"
                     "v_add_f32_dpp %5, %12, %12 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %13, %13 row_ror:4 row_mask:0xf bank_mask:0xa\n"
                     "v_add_f32_dpp %6, %14, %14 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %15, %15 row_ror:4 row_mask:0xf bank_mask:0xa\n"
                     "v_add_f32_dpp %7, %16, %16 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %17, %17 row_ror:4 row_mask:0xf bank_mask:0xa\n"
                     "v_add_f32_dpp %0, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %0, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xc\n"
                     "v_add_f32_dpp %1, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n"
                     "v_add_f32_dpp %2, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                     "s_nop 1"
                     : "=&v"(q03), "=&v"(q47), "=&v"(q8), "=&v"(r01), "=&v"(r23), "=&v"(r45), "=&v"(r67), "=&v"(r8)
                     : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v9));
    else
        asm volatile("s_nop 1\n"
                     "v_add_f32_dpp %3, %8, %8 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %9, %9 row_ror:4 row_mask:0xf bank_mask:0xa\n"
                     "v_add_f32_dpp %4, %10, %10 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %4, %11, %11 row_ror:4 row_mask:0xf bank_mask:0xa\n"
                     "v_add_f32_dpp %5, %12, %12 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %13, %13 row_ror:4 row_mask:0xf bank_mask:0xa\n"
                     "v_add_f32_dpp %6, %14, %14 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %15, %15 row_ror:4 row_mask:0xf bank_mask:0xa\n"
                     "v_add_f32_dpp %7, %16, %16 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                     "v_add_f32_dpp %0, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %0, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xc\n"
                     "v_add_f32_dpp %1, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n"
                     "v_add_f32_dpp %2, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                     "s_nop 1"
                     : "=&v"(q03), "=&v"(q47), "=&v"(q8), "=&v"(r01), "=&v"(r23), "=&v"(r45), "=&v"(r67), "=&v"(r8)
                     : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]));
#else
    q03 = q47 = q8 = 0.f; (void)v;
#endif
    const bool b0 = lane & 1, b1 = lane & 2;
    const float o07 = bfly_pair<0x4E>(q03, q47, b1);
    const float o8 = bfly_single<0x4E>(q8);
    float t = bfly_pair<0xB1>(o07, o8, b0);
    {
        const unsigned u = __float_as_uint(t);
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);   // rows (0,1) and (2,3) exchange
        t = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {
        const unsigned u = __float_as_uint(t);
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // halves exchange
        t = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    return t;
}

// sum over each 16-lane DPP row: total lands in lanes 15, 31, 47, 63
__device__ __forceinline__ float row_sum_to_lane15(float v)
{
    v = dpp_add<0x111, 0xf>(v);
    v = dpp_add<0x112, 0xf>(v);
    v = dpp_add<0x114, 0xf>(v);
    v = dpp_add<0x118, 0xf>(v);
    return v;
}

#ifndef GSR_PRE_THREADS
#define GSR_PRE_THREADS 128      // (experiment: 64 = one wave per block of the per-Gaussian kernels -- same waves per CU, barriers that cost nothing; the
#endif                           //  batched parameter store pads its models to 128: single renders only)
constexpr int kPreThreads = GSR_PRE_THREADS;
constexpr uint32_t kDepthKeyBias = 0x3E4CCCCDu;   // bit pattern of the near plane, 0.2f (gsr_math.h kNearZ): no visible Gaussian's depth key lies below it
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x);   // (defined with the direct binning)
constexpr int kShStride = 49;   // 48 floats + 1 pad: conflict-free column reads

// Coalesced 16-byte streaming of a block's contiguous [nG][ROW] float rows into / out of the padded LDS tile
// (row stride kShStride, first column `col0`).  Used when the rows are dense in memory (the stored row IS the
// active row), the block base is 16-byte aligned and nG*ROW is a multiple of 4; otherwise the scalar loops run.
template <int ROW>
__device__ __forceinline__ void stage_in_vec(float* s_sh, int col0, const float* __restrict__ src, int nG, int tid)
{
    const float4* s4 = reinterpret_cast<const float4*>(src);
    for (int v = tid; v < (nG * ROW) / 4; v += kPreThreads) {
        const float4 x = s4[v];
        const int f = 4 * v;
        int g = f / ROW, e = f - g * ROW;
        const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            s_sh[g * kShStride + col0 + e] = xs[q];
            if (++e == ROW) { e = 0; g++; }
        }
    }
}
template <int ROW>
__device__ __forceinline__ void stage_out_vec(const float* s_sh, int col0, float* __restrict__ dst, int nG, int tid)
{
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int v = tid; v < (nG * ROW) / 4; v += kPreThreads) {
        const int f = 4 * v;
        int g = f / ROW, e = f - g * ROW;
        float xs[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            xs[q] = s_sh[g * kShStride + col0 + e];
            if (++e == ROW) { e = 0; g++; }
        }
        d4[v] = make_float4(xs[0], xs[1], xs[2], xs[3]);
    }
}
// Linear tiles: with split dc / rest storage and every stored coefficient active, a block's rows are copied into LDS in
// their memory order (16-byte LDS accesses, no index arithmetic, no bank conflicts on the streaming side); the owning
// thread walks its row at stride ROW, which is conflict-free when ROW is odd (3 and 45).  The padded stride-49 tile above
// stays for every other layout.
template <int ROW>
__device__ __forceinline__ void stage_in_lin(float* tile, const float* __restrict__ src, int nG, int tid)
{
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* t4 = reinterpret_cast<float4*>(tile);
    for (int v = tid; v < (nG * ROW) / 4; v += kPreThreads) t4[v] = s4[v];
}
template <int ROW>
__device__ __forceinline__ void stage_out_lin(const float* tile, float* __restrict__ dst, int nG, int tid)
{
    const float4* t4 = reinterpret_cast<const float4*>(tile);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int v = tid; v < (nG * ROW) / 4; v += kPreThreads) d4[v] = t4[v];
}
__device__ __forceinline__ bool vec_ok(const void* p, int nfloats) { return (((uintptr_t)p & 15) == 0) && ((nfloats & 3) == 0); }

// stage_in_vec split in two: the 16-byte loads are issued into registers first, the per-Gaussian arithmetic that does
// not need the rows runs while they are in flight, and only then are they scattered into the padded LDS tile.
template <int ROW>
constexpr int stage_regs() { return (ROW * kPreThreads / 4 + kPreThreads - 1) / kPreThreads; }
template <int ROW, int KN>
__device__ __forceinline__ void stage_load(float4 (&r)[KN], const float* __restrict__ src, int nG, int tid)
{
    static_assert(KN >= stage_regs<ROW>(), "register file too small for the rows");
    const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll
    for (int q = 0; q < stage_regs<ROW>(); q++) {
        const int v = tid + q * kPreThreads;
        r[q] = v < (nG * ROW) / 4 ? s4[v] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int ROW, int KN>
__device__ __forceinline__ void stage_store(const float4 (&r)[KN], float* s_sh, int col0, int nG, int tid)
{
#pragma unroll
    for (int q = 0; q < stage_regs<ROW>(); q++) {
        const int v = tid + q * kPreThreads;
        if (v < (nG * ROW) / 4) {
            const int f = 4 * v;
            int g = f / ROW, e = f - g * ROW;
            const float xs[4] = {r[q].x, r[q].y, r[q].z, r[q].w};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                s_sh[g * kShStride + col0 + e] = xs[c];
                if (++e == ROW) { e = 0; g++; }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K1: per-Gaussian projection.  SH rows are staged through LDS with coalesced loads.
// ------------------------------------------------------------------------------------------------
// RAW = true ("next" row f-2): the kernel consumes the model's raw parameters and applies the activations of
// /root/reference/scene/gaussian_model_ht.py:49-65,128-133,176-188 itself -- scale = exp(_scaling),
template <int ROW, int KN>
__device__ __forceinline__ void stage_store_lin(const float4 (&r)[KN], float* tile, int nG, int tid)
{
    float4* t4 = reinterpret_cast<float4*>(tile);
#pragma unroll
    for (int q = 0; q < stage_regs<ROW>(); q++) {
        const int v = tid + q * kPreThreads;
        if (v < (nG * ROW) / 4) t4[v] = r[q];
    }
}

// whole blocks (nG == kPreThreads): the same two halves with compile-time bounds
template <int ROW, int KN>
__device__ __forceinline__ void stage_load_full(float4 (&r)[KN], const float* __restrict__ src, int tid)
{
    static_assert(KN >= stage_regs<ROW>(), "register file too small for the rows");
    constexpr int total4 = kPreThreads * ROW / 4;
    const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll
    for (int q = 0; q < stage_regs<ROW>(); q++) {
        const int v = tid + q * kPreThreads;
        if ((q + 1) * kPreThreads <= total4 || v < total4) r[q] = s4[v];
    }
}
template <int ROW, int KN>
__device__ __forceinline__ void stage_store_lin_full(const float4 (&r)[KN], float* tile, int tid)
{
    constexpr int total4 = kPreThreads * ROW / 4;
    float4* t4 = reinterpret_cast<float4*>(tile);
#pragma unroll
    for (int q = 0; q < stage_regs<ROW>(); q++) {
        const int v = tid + q * kPreThreads;
        if ((q + 1) * kPreThreads <= total4 || v < total4) t4[v] = r[q];
    }
}

// q = normalize(_rotation), opacity = sigmoid(_opacity), SH = cat(_features_dc, _features_rest) -- so the
// torch exp / sigmoid / normalize / cat kernels (and their backward) disappear from the train step.
// Large rects (more than 32 candidate tiles) are counted by the whole wave, 64 candidate tiles per step, so one
// screen-filling splat costs its wave rect/64 steps instead of rect steps of a single lane.  Every lane of the wave must
// call this (it ballots and shuffles).  Shared by k_preprocess and the next-view tail of k_preprocess_bwd.
__device__ __forceinline__ void count_large_rects(bool act, Splat& s, TileRec& rec, int W, int H, int tiles_x, int tiles_y, int tid)
{
    const int lane = tid & 63;
    for (unsigned long long pm = __ballot(act && s.tiles == kTilesPending); pm != 0ull; pm &= pm - 1ull) {
        const int src = (int)__builtin_ctzll(pm);
        const float bpx = __shfl(s.px, src, 64), bpy = __shfl(s.py, src, 64), bca = __shfl(s.ca, src, 64);
        const float bcb = __shfl(s.cb, src, 64), bcc = __shfl(s.cc, src, 64), bop = __shfl(s.op, src, 64);
        const int brad = __shfl(s.radius, src, 64);
        int x0, y0, x1, y1;
        tile_rect_tight(bpx, bpy, brad, bca, bcb, bcc, bop, W, H, tiles_x, tiles_y, x0, y0, x1, y1);
        const int ww = x1 - x0, full = ww * (y1 - y0);
        const TileTest tt = make_tile_test(bpx, bpy, bca, bcb, bcc, bop);
        uint32_t cnt = 0;
        for (int c0 = 0; c0 < full; c0 += 64) {
            const int c = c0 + lane;
            bool ok = false;
            if (c < full) {
                const int ty = c / ww, tx = c - ty * ww;
                ok = tile_accept(tt, x0 + tx, y0 + ty, W, H);
            }
            cnt += (uint32_t)__popcll(__ballot(ok));
        }
        if (lane == src) { s.tiles = cnt; rec.mask = cnt; }
    }
}

// The producer of the depth keys can count the depth sort's digits (4 x 8 bits, per run: radix_sort.h onesweep_run_len) and
// clear that sort's status words, so the sort needs no histogram launch (onesweep_sort_pairs hist_done).  ghist == nullptr: off.
// The counters must be zero before the kernel starts.
struct DepthHist {
    uint32_t* ghist;
    uint32_t* status;
    uint32_t status_words;
    uint32_t clear_threads;   // threads of k_preprocess_bwd that run its next-view tail (a ragged last block does not): the stride of
                              // their status clear; 0 = none do, the single-block k_preprocess launch clears instead
};

template <int DEG, bool RAW>
__global__ __launch_bounds__(kPreThreads) void k_preprocess(CamParams cp, int N, const float* __restrict__ means,
                                                            const float* __restrict__ scales, const float* __restrict__ rots,
                                                            const float* __restrict__ cov_pre, const float* __restrict__ opac,
                                                            const float* __restrict__ shs, const float* __restrict__ shs_rest,
                                                            const float* __restrict__ colors,
                                                            Splat* __restrict__ splat, int32_t* __restrict__ radii,
                                                            uint32_t* __restrict__ dkey, uint32_t* __restrict__ gid,
                                                            TileRec* __restrict__ tilerec, uint32_t* __restrict__ zero_words,
                                                            int zero_count, int block0, DepthHist dh, uint2* __restrict__ early_parts = nullptr,
                                                            uint32_t window_max = 0xffffffffu, uint8_t* __restrict__ visible = nullptr)
{
    constexpr int NC3 = 3 * (DEG + 1) * (DEG + 1);
    __shared__ float s_sh[kPreThreads * kShStride];
    __shared__ uint32_t s_early[2][kPreThreads / 64];
    const int tid = threadIdx.x;
    const int base = ((int)blockIdx.x + block0) * kPreThreads;   // block0: first block of a partial launch (0 = the whole cloud)
    const int i = base + tid;
    const int model = select_view(cp, (int)blockIdx.x + block0);   // batched render: this block's model and its camera
    if (blockIdx.x == 0)   // rides along: clear the head of the depth sort's scratch (saves a fill launch)
        for (int q = tid; q < zero_count; q += kPreThreads) zero_words[q] = 0u;
    // ---- phase 0: SH rows.  Dense, aligned rows (the normal case) are only LOADED here -- into registers; the geometry
    // below does not need them, so the 180-192 bytes per Gaussian stream in underneath ~1500 instructions of float64
    // projection instead of in front of them (K1 was the sum of its HBM time and its VALU time: 0.073 ms without
    // SH rows, 0.125 ms with).  Odd layouts (partial last block, stored degree above the active one) stage directly.
    constexpr int NR = NC3 > 3 ? NC3 - 3 : 1;
    float4 r_dc[1];                        // 128 x 3 floats = 96 float4
    float4 r_rows[stage_regs<NC3>()];      // the rest rows OR the full rows (one register file for either layout)
    bool pend_dc = false, pend_rest = false, pend_full = false;
    const int nG = min(kPreThreads, N - base);
    if (shs) {
        if (shs_rest) {   // split storage: dc [N,1,3] + rest [N,M-1,3]; two straight, divergence-free streams
            const size_t row = (size_t)(cp.M - 1) * 3;
            const float* dc0 = shs + (size_t)base * 3;
            if (vec_ok(dc0, nG * 3)) { stage_load<3>(r_dc, dc0, nG, tid); pend_dc = true; }
            else for (int f = tid; f < nG * 3; f += kPreThreads) s_sh[(f / 3) * kShStride + (f % 3)] = dc0[f];
            if (NC3 > 3) {
                const float* r0 = shs_rest + (size_t)base * row;
                if (row == NR && vec_ok(r0, nG * NR)) { stage_load<NR>(r_rows, r0, nG, tid); pend_rest = true; }
                else for (int f = tid; f < nG * NR; f += kPreThreads) {
                    const int g = f / NR, e = f - g * NR;
                    s_sh[g * kShStride + 3 + e] = shs_rest[(size_t)(base + g) * row + e];
                }
            }
        } else {
            const size_t row = (size_t)cp.M * 3;
            const float* r0 = shs + (size_t)base * row;
            if (row == NC3 && vec_ok(r0, nG * NC3)) { stage_load<NC3>(r_rows, r0, nG, tid); pend_full = true; }
            else for (int f = tid; f < nG * NC3; f += kPreThreads) {
                const int g = f / NC3, e = f - g * NC3;
                s_sh[g * kShStride + e] = shs[(size_t)(base + g) * row + e];
            }
        }
    }
    // ---- phase 1: geometry (+ precomputed colour), no SH
    const bool act = i < N;
    Camera cam = load_camera(cp);
    cam.D = DEG;
    float mean[3] = {0.f, 0.f, 0.f};
    Splat s;
    TileRec rec;
    uint32_t lo_pack = 0u;   // sub-ulp remainders of the pixel-space mean (gsr_math.h pixel_lo_pack): stored in the record's `tiles` word
    if (act) {
        mean[0] = means[3 * (size_t)i]; mean[1] = means[3 * (size_t)i + 1]; mean[2] = means[3 * (size_t)i + 2];
        apply_points_transform(cp.xf, mean);
        float sc[3] = {0, 0, 0}, rq[4] = {1, 0, 0, 0}, cv[6], colp[3];
        if (cov_pre) {
#pragma unroll
            for (int k = 0; k < 6; k++) cv[k] = cov_pre[6 * (size_t)i + k];
        } else {
#pragma unroll
            for (int k = 0; k < 3; k++) sc[k] = scales[3 * (size_t)i + k];
#pragma unroll
            for (int k = 0; k < 4; k++) rq[k] = rots[4 * (size_t)i + k];
        }
        if (colors) {
#pragma unroll
            for (int k = 0; k < 3; k++) colp[k] = colors[3 * (size_t)i + k];
        }
        float op = opac[i];
        if (RAW) {
            if (!cov_pre) {
#pragma unroll
                for (int k = 0; k < 3; k++) sc[k] = expf(sc[k]);
                const float inv = 1.0f / fmaxf(sqrtf(rq[0] * rq[0] + rq[1] * rq[1] + rq[2] * rq[2] + rq[3] * rq[3]), 1e-12f);
#pragma unroll
                for (int k = 0; k < 4; k++) rq[k] *= inv;
            }
            op = 1.0f / (1.0f + expf(-op));
        }
        preprocess_one(cam, mean, sc, rq, cov_pre ? cv : nullptr, op, nullptr, 3, 1, colors ? colp : nullptr, s, &rec, true, &lo_pack);
    }
    count_large_rects(act, s, rec, cp.W, cp.H, cam.tiles_x, cam.tiles_y, tid);
    rec.rect += (uint32_t)(model * cam.tiles_y) << 12;   // tile rows of model b start at b * tiles_y in the batch's tall tile grid
    // ---- early R (round 5): the instance count is the sum of the tile counts, whatever order the depth sort puts them in -- this
    // block's share (and its count of visible depth keys beyond the three-pass sort's window) goes to early_parts[block]; the
    // depth sort's histogram kernel adds the shares up and hands the total to the host (radix_sort.h OsRider) while the sort,
    // the tile counts and the scan are still to run
    if (early_parts) {
        const uint32_t nt = act ? s.tiles : 0u;
        const uint32_t far = (nt > 0u && __float_as_uint(s.depth) - kDepthKeyBias > window_max) ? 1u : 0u;
        const uint32_t st_ = wave_inclusive_sum(nt), sf_ = wave_inclusive_sum(far);
        if ((tid & 63) == 63) { s_early[0][tid >> 6] = st_; s_early[1][tid >> 6] = sf_; }
        __syncthreads();
        if (tid == 0) {
            uint32_t t0 = 0u, t1 = 0u;
#pragma unroll
            for (int w = 0; w < kPreThreads / 64; w++) { t0 += s_early[0][w]; t1 += s_early[1][w]; }
            early_parts[blockIdx.x + block0] = make_uint2(t0, t1);
        }
    }
    // ---- phase 2: rows into the LDS tile, then the colour of the Gaussians that survived the culls
    if (shs) {
        constexpr bool kLinOk = NC3 > 3 && (NR & 1);
        const bool lin = kLinOk && pend_dc && pend_rest;   // block-uniform: linear tiles (see stage_in_lin)
        float* s_dc = s_sh;
        float* s_rest = s_sh + kPreThreads * 3;
        if (lin) {
            stage_store_lin<3>(r_dc, s_dc, nG, tid);
            stage_store_lin<NR>(r_rows, s_rest, nG, tid);
        } else {
            if (pend_dc) stage_store<3>(r_dc, s_sh, 0, nG, tid);
            if (pend_rest) stage_store<NR>(r_rows, s_sh, 3, nG, tid);
            if (pend_full) stage_store<NC3>(r_rows, s_sh, 0, nG, tid);
        }
        __syncthreads();
        if (act && s.radius > 0) {
            float col[3];
            if (lin) splat_sh_color(cam, mean, s_rest + tid * NR - 3, 3, 1, col, s_dc + tid * 3);
            else splat_sh_color(cam, mean, &s_sh[tid * kShStride], 3, 1, col);
            s.r = col[0]; s.g = col[1]; s.b = col[2];
        }
    }
    if (dh.ghist && dh.clear_threads == 0u)   // fewer than 128 Gaussians, ragged: no block of k_preprocess_bwd ran the next-view tail
        for (uint32_t q = tid; q < dh.status_words / 4; q += kPreThreads) reinterpret_cast<uint4*>(dh.status)[q] = make_uint4(0u, 0u, 0u, 0u);
    if (!act) return;
    {   // the stored record carries the remainders where the tile count sat (nobody reads the count back; it lives on in registers below)
        Splat st = s;
        st.tiles = lo_pack;
        splat[i] = st;
    }
    tilerec[i] = rec;   // compact emission record: k_tile_counts / k_emit never touch the 48 B splats
    radii[i] = s.radius;
    if (visible) visible[i] = s.radius > 0 ? 1 : 0;   // `visibility_filter = radii > 0` (gaussian_model_ht.py:905), written here instead of by a torch launch
    const uint32_t key = s.tiles > 0 ? __float_as_uint(s.depth) : 0xffffffffu;
    dkey[i] = key;
    gid[i] = (uint32_t)i;
    if (dh.ghist) {   // (only the single ragged block behind k_preprocess_bwd's next-view tail comes here: plain global adds)
        const uint32_t x = (uint32_t)i / onesweep_run_len<uint32_t>((uint32_t)N);
#pragma unroll
        for (int p = 0; p < 4; p++) atomicAdd(&dh.ghist[((size_t)p * kOsRanges + x) * 256 + ((key >> (8 * p)) & 255u)], 1u);
    }
}

// first kernel of a forward that received a prepared buffer: radii to the caller's tensor + the clears k_preprocess's block 0 does
__global__ __launch_bounds__(256) void k_prepared_begin(int N, const int32_t* __restrict__ radii_in, int32_t* __restrict__ radii_out,
                                                        uint32_t* __restrict__ zero_words, int zero_count)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < N) radii_out[i] = radii_in[i];
    if (blockIdx.x == 0)
        for (int q = threadIdx.x; q < zero_count; q += 256) zero_words[q] = 0u;
}

__global__ void k_mark_visible(int N, const float* __restrict__ means, const float* __restrict__ vm, uint8_t* __restrict__ present)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = vm[k];
    present[i] = depth_key(v, means[3 * (size_t)i], means[3 * (size_t)i + 1], means[3 * (size_t)i + 2]) > kNearZ ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// K3: tiles-touched in depth order -> per-block sums -> exclusive block offsets + total
// ------------------------------------------------------------------------------------------------
constexpr int kEmitThreads = 256;

__device__ __forceinline__ uint32_t block_inclusive_scan_256(uint32_t v, uint32_t* s_wave /*[4]*/, uint32_t& total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();
    uint32_t add = 0;
    for (int w = 0; w < wave; w++) add += s_wave[w];
    total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    return x + add;
}

// small clears that ride in k_block_scan (4-byte words; see gsr_forward)
struct ZeroJobs {
    void* p[3];
    uint32_t words[3];
};

// one workgroup of NTHR threads: exclusive scan of block_sums[nb] in place; total (64-bit) -> *total_out and, when
// host_out is given, into pinned host memory (device-visible): the count reaches the host without a copy launch
// late_out (round 6): with early R the host has read its count -- and the sorts' give-up counter -- from the rider on the
// depth sort's FIRST kernel, before this forward's look-backs ran.  The scan still reports what it saw, into spare words of the same
// pinned slot: {total, sequence, give-ups} at late_out[0..2]; the host compares them with the early values the next time it takes the
// slot (or enters gsr_backward) and fails loudly on a mismatch (late_check).
__device__ __forceinline__ void scan_report_late(unsigned long long* late_out, unsigned long long all, unsigned long long seq)
{
    __atomic_store_n(late_out, all, __ATOMIC_RELAXED);
    __atomic_store_n(late_out + 2, (unsigned long long)g_onesweep_giveups, __ATOMIC_RELAXED);
    __threadfence_system();
    __atomic_store_n(late_out + 1, seq, __ATOMIC_RELAXED);
    __threadfence_system();
}
template <int NTHR>
__device__ __forceinline__ void block_scan_body(uint32_t* block_sums, int nb, unsigned long long* total_out, const ZeroJobs& zj,
                                                unsigned long long* host_out, unsigned long long host_seq, unsigned long long* s_wsum /*[NTHR/64]*/,
                                                const unsigned int* window_overflow = nullptr, unsigned long long* late_out = nullptr)
{
    constexpr int NWV = NTHR / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        uint32_t* z = static_cast<uint32_t*>(zj.p[j]);
        for (uint32_t q = tid; q < zj.words[j]; q += NTHR) z[q] = 0u;
    }
    const int chunk = (nb + NTHR - 1) / NTHR;
    const int lo = min(nb, tid * chunk), hi = min(nb, lo + chunk);
    unsigned long long sum = 0;
    for (int b = lo; b < hi; b++) sum += block_sums[b];
    // inclusive scan over the NTHR partial sums: wave shuffles, then the wave totals (one barrier)
    unsigned long long inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long y = __shfl_up(inc, off, 64);
        if (lane >= off) inc += y;
    }
    if (lane == 63) s_wsum[wave] = inc;
    __syncthreads();
    unsigned long long add = 0, all = 0;
#pragma unroll
    for (int w = 0; w < NWV; w++) { const unsigned long long t = s_wsum[w]; add += (w < wave) ? t : 0ull; all += t; }
    unsigned long long run = add + inc - sum;
    for (int b = lo; b < hi; b++) {
        const uint32_t c = block_sums[b];
        block_sums[b] = (uint32_t)run;
        run += c;
    }
    if (tid == 0) {
        *total_out = all;
        if (host_out) {   // the count, then the call's sequence number: the host polls the second word (gsr_forward)
            __atomic_store_n(host_out + 2, (unsigned long long)g_onesweep_giveups, __ATOMIC_RELAXED);   // (sorts that gave up so far: see radix_sort.h)
            __atomic_store_n(host_out + 3, (unsigned long long)(window_overflow ? *window_overflow : 0u), __ATOMIC_RELAXED);   // (depth keys beyond the 9-bit sort's window)
            __atomic_store_n(host_out, all, __ATOMIC_RELAXED);
            __threadfence_system();
            __atomic_store_n(host_out + 1, host_seq, __ATOMIC_RELAXED);
            __threadfence_system();
        }
        if (late_out) scan_report_late(late_out, all, host_seq);
    }
}

// ------------------------------------------------------------------------------------------------
// Balanced placement of the forward blend's waves (round 4).  Measured (tools/k7_slot_map.py): within an XCD the dispatcher deals
// single-wave workgroups to its 128 SIMDs strictly round-robin -- every aligned run of 128 consecutive slots covers each SIMD once,
// slot k and slot k + 128 share a SIMD (which SIMD that is rotates from launch to launch) -- and a SIMD finishes at
// 0.0376 us x (the visits of its 8-9 waves) + 20 us.  The kernel therefore lasts as long as its most loaded COLUMN k mod 128.  A
// wave's visits are predictable: the same view is rendered again a few iterations later (a trainer cycles through its frames) with
// nearly the same scene (correlation 0.95-0.96 at eight steps' distance; 0.47 with the previous step's OTHER view).  So: the blend
// records every wave's visits in a small device-side cache keyed by a hash of the view matrix; the next render of that view sorts
// each XCD's items by that prediction (descending) and deals them to the slots in snake order -- column totals even out, the
// shortest items are the ones that start late -- in eight extra workgroups of k_tile_counts, off the critical path.  Affinity
// for speed only: the permutation is a permutation whatever the cache holds (a miss, a race with another stream: identity or a
// poorer balance, never a different image).
// Measured and dropped (round 4): CHAINING the items beyond the 1 024 waves an XCD holds (96 of 1 120 slots on the 980x545 frame, 59 of
// them empty edge sub-tiles or padding) behind the shortest resident items, so that nothing starts late -- the wave loop needs the
// kernel arguments re-read per item to stay at 8 waves per SIMD (+4 us on its own), and a chained pair pays a wave's fixed ~19 us
// start twice IN SERIES on one wave where the dispatcher starts the late item on the first slot that frees anywhere in the XCD:
// 116 us against 98 (identity order: 111).
// ------------------------------------------------------------------------------------------------
constexpr int kVcEntries = 128;
constexpr int kVcPoseFloats = 28;   // view matrix (16) + points_transform (12; identity when the caller passes none)
struct ViewCostHdr {
    unsigned long long key[kVcEntries];     // 0 = empty; 2 = keyed by pose (nearest stored pose within the tolerance); odd = keyed by the caller's view id
    uint32_t stamp[kVcEntries];
    uint32_t clock, pad[31];
    float pose[kVcEntries][kVcPoseFloats];  // the pose of the entry's last render (both kinds of key)
};
struct BlendBalance {       // kernel-argument bundle; hdr == nullptr: off
    ViewCostHdr* hdr;
    uint16_t* cost;         // [kVcEntries][items]: visits of workgroup-item (xcd + 8 kslot) at the last render of that view
    uint16_t* perm;         // per call: dispatch slot (xcd + 8 k) -> item's kslot
    uint32_t* cur;          // per call: [0] cache entry this render records into (0xffffffff: none), [1 + x] XCD x's slice of perm is valid
    const float* vm;        // what identifies the frame: the view matrix (16 floats) ...
    const float* pt;        // ... and the points transform (12 floats, or nullptr) -- the reference trains with an identity camera and moves the points
    long long view_id;      // ... or, when non-zero, the caller's frame id (GsrForwardArgs::view_id): exact match, the pose may drift freely
    float tol;              // pose match: largest absolute difference of any of the 28 entries
    int items, nslots4, W, H;
};

// Which cache entry is this render's view?  (round 5: until round 4 the key was a bit-exact hash of the view matrix -- under the
// reference's calling convention, identity camera + the pose through get_xyz (/root/reference/trainer/trainer.py:993-995,
// scene/gaussian_model_ht.py:135-148), every frame hashed to ONE entry; and a camera that carries a pose under refinement changes its
// bits every step (ht3dgs_trainer.py:162-166) and never hit.)  With a view id: the entry of that id.  Without: the entry whose stored
// pose (view matrix and points transform) is nearest to this render's, if within `tol` in every entry -- a pose under refinement moves
// by ~1e-4..1e-3 per step and the entry follows it (the stored pose is replaced on every hit); two frames closer than `tol` to each other
// share an entry, which is harmless: their visit counts are as correlated as one frame's with itself a few steps later.
// Every builder workgroup decides for itself (they cannot synchronise); the decisions may differ when another stream touches the cache
// in between -- or when workgroup 0 of THIS launch, which keeps the books, has already rewritten the entry's key / pose while another
// builder is still reading them (an unordered read of words that only ever hold a former or a new pose: the reader takes the entry or
// misses it) -- each XCD's slice of perm carries its own valid flag, so any mix is still a permutation per XCD: speed, never the image.
// (The header and the cost tables are cleared synchronously when they are allocated, so no word of them is ever uninitialised.)
__device__ __forceinline__ unsigned long long view_id_key(long long id)
{
    unsigned long long h = 1469598103934665603ull ^ (unsigned long long)id;
    h *= 1099511628211ull; h ^= h >> 29; h *= 1099511628211ull;
    return h | 1ull;
}

__device__ void balance_build(const BlendBalance bb, const int x)
{
    __shared__ uint32_t s_hist[1024];
    __shared__ unsigned long long s_best;
    __shared__ float s_pose[kVcPoseFloats];
    const int tid = threadIdx.x;
    if (tid == 0) s_best = ~0ull;
    if (tid < 16) s_pose[tid] = bb.vm[tid];
    else if (tid < kVcPoseFloats) s_pose[tid] = bb.pt ? bb.pt[tid - 16] : (((tid - 16) % 5 == 0) ? 1.f : 0.f);   // rows of [I | 0]
    for (int q = tid; q < 1024; q += kEmitThreads) s_hist[q] = 0u;
    __syncthreads();
    const unsigned long long want = bb.view_id ? view_id_key(bb.view_id) : 2ull;
    if (tid < kVcEntries && bb.hdr->key[tid] == want) {
        float d = 0.f;
        if (!bb.view_id) {
            const float* p = bb.hdr->pose[tid];
            bool near = true;        // (fmaxf drops a NaN difference: a NaN pose must match nothing)
            for (int i = 0; i < kVcPoseFloats; i++) { const float df = fabsf(p[i] - s_pose[i]); near = near && (df <= bb.tol); d = fmaxf(d, df); }
            if (!near) d = 3.0e38f;
        }
        if (d <= bb.tol) atomicMin(&s_best, ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)tid);
    }
    __syncthreads();
    const int e = s_best == ~0ull ? -1 : (int)(s_best & 0xffffffffull);
    if (x == 0 && tid == 0) {   // workgroup 0 keeps the cache's books: touch the entry, or take the least recently used one
        int rec = e;
        const uint32_t now = bb.hdr->clock + 1u;
        if (rec < 0) {
            uint32_t best = 0xffffffffu;
            for (int i = 0; i < kVcEntries; i++) if (bb.hdr->stamp[i] < best) { best = bb.hdr->stamp[i]; rec = i; }
            bb.hdr->key[rec] = want;
        }
        for (int i = 0; i < kVcPoseFloats; i++) bb.hdr->pose[rec][i] = s_pose[i];
        bb.hdr->stamp[rec] = now; bb.hdr->clock = now;
        bb.hdr->pad[0] += 1u; bb.hdr->pad[1] += e >= 0 ? 1u : 0u;      // lookups / hits (gsr_debug_view_cache_stats)
        bb.cur[0] = (uint32_t)rec;
        bb.cur[9] = e >= 0 ? 1u : 0u;      // (the list cut asks builder 0 alone: the keeper of the books decides which entry's keys count)
    }
    if (tid == 0) bb.cur[1 + x] = e >= 0 ? 1u : 0u;
    if (e < 0) return;          // first render of this view: identity placement (the blend ignores perm)
    const uint16_t* cost = bb.cost + (size_t)e * bb.items;
    // counting sort of this XCD's items by predicted visits, descending; ties in any order
    constexpr int kPer = 20;    // items per thread: up to 5 120 per XCD (65 535 workgroups in all)
    uint32_t c[kPer];
#pragma unroll
    for (int q = 0; q < kPer; q++) {
        const int k = tid + q * kEmitThreads;
        c[q] = k < bb.nslots4 ? min((uint32_t)cost[x + 8 * k], 1023u) : 0xffffffffu;
        if (k < bb.nslots4) atomicAdd(&s_hist[1023u - c[q]], 1u);
    }
    __syncthreads();
    // exclusive scan of the 1024 bins (bin 0 = the largest cost): four bins per thread
    __shared__ uint32_t s_wave[4];
    uint32_t v4[4], run = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) { v4[q] = s_hist[4 * tid + q]; run += v4[q]; }
    uint32_t total;
    uint32_t base = block_inclusive_scan_256(run, s_wave, total) - run;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) { s_hist[4 * tid + q] = base; base += v4[q]; }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kPer; q++) {
        const int k = tid + q * kEmitThreads;
        if (k >= bb.nslots4) continue;
        const uint32_t r = atomicAdd(&s_hist[1023u - c[q]], 1u);    // rank in descending order
        const uint32_t round = r >> 7, col = r & 127u;
        const uint32_t pos = round * 128u + ((round & 1u) ? 127u - col : col);      // snake over the 128 columns
        // (the last, partial round keeps its positions inside the range: mirror within what is left of it)
        const uint32_t left = (uint32_t)bb.nslots4 - round * 128u;
        const uint32_t pos2 = left >= 128u ? pos : round * 128u + ((round & 1u) ? left - 1u - col : col);
        bb.perm[x + 8 * (int)pos2] = (uint16_t)k;
    }
}

// sorted_gid == nullptr (the tile-sort route on the emit path, round 5): index order -- no depth sort in front, the records are counted
// where they lie (no gather, no sorted copy), and the launch's last workgroup carries the rider that publishes R.
__global__ __launch_bounds__(kEmitThreads) void k_tile_counts(int N, const uint32_t* __restrict__ sorted_gid,
                                                              const TileRec* __restrict__ tilerec, uint32_t* __restrict__ block_sums,
                                                              TileRec* __restrict__ sorted_rec, BlendBalance bb, int nb, OsRider rider = OsRider{})
{
    // (eight extra workgroups when bb.hdr is set -- the FIRST eight, so that they run beside the counting, not behind it)
    const int nbuild = bb.hdr ? 8 : 0, blk = (int)blockIdx.x - nbuild;
    if (blk < 0) { balance_build(bb, (int)blockIdx.x); return; }
    (void)nb;
    if (rider.host && blockIdx.x == gridDim.x - 1) {   // (block-uniform)
        __shared__ unsigned long long s_r[2][16];
        onesweep_rider_publish(rider, s_r);
    }
    __shared__ uint32_t s_wave[4];
    const int j = blk * kEmitThreads + threadIdx.x;
    uint32_t t = 0;
    if (j < N) {   // the one random gather of the records: k_emit reads them back in depth order, coalesced
        const TileRec r = tilerec[sorted_gid ? sorted_gid[j] : (uint32_t)j];
        if (sorted_rec) sorted_rec[j] = r;
        t = tilerec_count(r);
    }
    uint32_t total;
    block_inclusive_scan_256(t, s_wave, total);
    if (threadIdx.x == 0) block_sums[blk] = total;
}

// (letting the LAST workgroup of k_tile_counts do the scan -- a ticket, agent-scope loads of the other blocks' totals --
//  saves the launch but measured 140 us at 1 M Gaussians and +2 us at 50 k: the totals of 3 907 blocks read through
//  coherent loads by 256 threads are a long latency chain; the single-workgroup launch below stays)
__global__ __launch_bounds__(1024) void k_block_scan(uint32_t* __restrict__ block_sums, int nb, unsigned long long* __restrict__ total_out,
                                                     ZeroJobs zj, unsigned long long* __restrict__ host_out, unsigned long long host_seq,
                                                     const unsigned int* __restrict__ window_overflow, unsigned long long* __restrict__ late_out = nullptr)
{
    __shared__ unsigned long long s_wsum[16];
    block_scan_body<1024>(block_sums, nb, total_out, zj, host_out, host_seq, s_wsum, window_overflow, late_out);
}

// ------------------------------------------------------------------------------------------------
// K4: emit (tile, gid) instances in depth order.  A block owns 256 depth-sorted Gaussians; its output base is the
// exclusive scan of the exact per-Gaussian tile counts.  The decisions of the exact tile test were recorded by
// k_preprocess as a bit mask over the rect (TileRec), so every output slot is a hit: lanes take consecutive output
// slots, find their Gaussian by binary search in the block's LDS scan and the tile by selecting the k-th set bit --
// no second evaluation of the test, no compaction, no 48-byte splat gather, consecutive addresses stored.
// Gaussians whose rect exceeds 32 tiles carry no mask: those are emitted one per wave with the test evaluated 64
// candidates at a time and a ballot compaction (same order: row-major over the rect).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int select_kth_bit(uint32_t m, uint32_t k)   // position of the k-th (0-based) set bit
{
    for (uint32_t i = 0; i < k; i++) m &= m - 1u;
    return __ffs((int)m) - 1;
}

// KeyT = uint16_t up to 65536 tiles (the tile sort then moves 6-byte records), uint32_t above
// EmitHist: the emission also counts the digits of the tile keys it writes, per run of the tile sort, and clears that sort's
// status words -- what k_radix_ghist would do in a launch of its own right behind it (radix_sort.h, onesweep_run_len).  A
// workgroup's outputs are consecutive, so they fall into at most two runs unless the frame is tiny: two LDS tables, direct
// global adds beyond.  ghist == nullptr: off.
struct EmitHist {
    uint32_t* ghist;
    uint32_t* status;
    uint32_t status_words;
    int bits;
    const unsigned long long* n_dev;
};

template <typename KeyT>
__global__ __launch_bounds__(kEmitThreads) void k_emit(int N, int W, int H, int tiles_x, int tiles_y,
                                                       const uint32_t* __restrict__ sorted_gid, const Splat* __restrict__ splat,
                                                       const TileRec* __restrict__ tilerec,
                                                       const uint32_t* __restrict__ block_offsets, KeyT* __restrict__ out_tile,
                                                       uint32_t* __restrict__ out_gid, uint32_t cap, EmitHist hz)
{
    __shared__ uint32_t s_hist[2][4][256];
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_incl[kEmitThreads];
    __shared__ uint32_t s_gid[kEmitThreads];
    __shared__ TileRec s_rec[kEmitThreads];
    __shared__ uint32_t s_big[kEmitThreads];
    __shared__ uint32_t s_nbig;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = blockIdx.x * kEmitThreads + tid;
    if (tid == 0) s_nbig = 0u;
    const bool hist = hz.ghist != nullptr;
    if (hist) {
#pragma unroll
        for (int q = 0; q < 8; q++) (&s_hist[0][0][0])[q * kEmitThreads + tid] = 0u;
        for (uint32_t q = blockIdx.x * kEmitThreads + tid; q < hz.status_words / 4; q += gridDim.x * kEmitThreads)
            reinterpret_cast<uint4*>(hz.status)[q] = make_uint4(0u, 0u, 0u, 0u);
    }
    uint32_t g = 0;
    TileRec r;
    r.mask = 0u; r.rect = 1u << 24;
    if (j < N) {
        g = sorted_gid ? sorted_gid[j] : (uint32_t)j;   // (nullptr: index order -- the tile-sort route)
        r = tilerec[j];   // already in depth order (k_tile_counts)
    }
    const uint32_t cnt = tilerec_count(r);
    uint32_t total;
    const uint32_t incl = block_inclusive_scan_256(cnt, s_wave, total);   // (has a barrier: s_nbig is visible)
    s_incl[tid] = incl; s_gid[tid] = g; s_rec[tid] = r;
    if ((r.rect & kTileRecBig) && cnt) s_big[atomicAdd(&s_nbig, 1u)] = (uint32_t)tid;   // order irrelevant: positions are absolute
    __syncthreads();
    const uint32_t base = block_offsets[blockIdx.x];
    // digit counting (hist): run geometry of the sort over n = min(cap, R) keys
    const int h_passes = onesweep_passes(hz.bits), h_dbits = onesweep_dbits(hz.bits);
    uint32_t run_len = 0, x0 = 0;
    unsigned long long b1 = ~0ull, b2 = ~0ull;
    if (hist) {
        const unsigned long long n_keys = hz.n_dev ? min((unsigned long long)cap, *hz.n_dev) : (unsigned long long)cap;
        run_len = onesweep_run_len<KeyT>((uint32_t)n_keys);
        if (run_len) { x0 = base / run_len; b1 = (unsigned long long)(x0 + 1) * run_len; b2 = b1 + run_len; }
    }
    auto count_key = [&](uint32_t key, uint32_t o) {
        const int t = o >= b1 ? (o >= b2 ? 2 : 1) : 0;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            if (p >= h_passes) break;
            const int wbits = min(h_dbits, hz.bits - h_dbits * p);
            const uint32_t d = (key >> (h_dbits * p)) & ((1u << wbits) - 1u);
            if (t < 2) atomicAdd(&s_hist[t][p][d], 1u);
            else atomicAdd(&hz.ghist[((size_t)p * kOsRanges + o / run_len) * 256 + d], 1u);
        }
    };
    // Every thread emits ITS Gaussian's tiles, in mask order (row-major over the rect), at base + its exclusive prefix.  (Round 3:
    // until then the lanes took consecutive OUTPUT slots and found their Gaussian by an eight-step binary search in LDS, the tile
    // by clearing k bits and a division -- coalesced stores, but ~150 instructions and eight dependent LDS reads per output; here a
    // lane's outputs are consecutive addresses of its own, neighbours' runs adjoin, and L2 merges the partial lines.)
    if (!(r.rect & kTileRecBig) && cnt) {
        const uint32_t ww = (r.rect >> 24) & 63u, rx0 = r.rect & 0xfffu, ry0 = (r.rect >> 12) & 0xfffu;
        const uint32_t rcp = (65536u + ww - 1u) / ww;   // floor(pos / ww) = pos * rcp >> 16 for pos < 32 <= 65536 / ww
        uint32_t o = base + incl - cnt;
        for (uint32_t m = r.mask; m != 0u; m &= m - 1u, o++) {
            const uint32_t pos = (uint32_t)__builtin_ctz(m), ty = (pos * rcp) >> 16, tx = pos - ty * ww;
            if (o < cap) {   // cap = R, or the speculative capacity of gsr_forward (then an overflow is re-run)
                const uint32_t key = (ry0 + ty) * (uint32_t)tiles_x + rx0 + tx;
                out_tile[o] = (KeyT)key;
                out_gid[o] = g;
                if (hist) count_key(key, o);
            }
        }
    }
    // large rects: one wave per Gaussian, the exact test on 64 candidate tiles at a time, survivors compacted in order
    const uint32_t nbig = s_nbig;
    const unsigned long long lt = lanemask_lt();
    for (uint32_t bi = wave; bi < nbig; bi += kEmitThreads / 64) {
        const int idx = (int)s_big[bi];
        const uint32_t gg = s_gid[idx];
        const Splat s = splat[gg];
        int x0, y0, x1, y1;
        tile_rect_tight(s.px, s.py, s.radius, s.ca, s.cb, s.cc, s.op, W, H, tiles_x, tiles_y, x0, y0, x1, y1);   // as k_preprocess
        const int vrow0 = (int)((s_rec[idx].rect >> 12) & 0xfffu) - y0;   // batched render: first tile row of this Gaussian's model (0 otherwise)
        const int ww = x1 - x0, full = ww * (y1 - y0);
        const TileTest tt = make_tile_test(s.px, s.py, s.ca, s.cb, s.cc, s.op);
        uint32_t run = base + (idx ? s_incl[idx - 1] : 0u);
        for (int c0 = 0; c0 < full; c0 += 64) {
            const int c = c0 + lane;
            bool ok = false;
            int gx = 0, gy = 0;
            if (c < full) {
                const int ty = c / ww, tx = c - ty * ww;
                gx = x0 + tx; gy = y0 + ty;
                ok = tile_accept(tt, gx, gy, W, H);
            }
            const unsigned long long m = __ballot(ok);
            if (ok) {
                const uint32_t o = run + (uint32_t)__popcll(m & lt);
                if (o < cap) {
                    const uint32_t key = (uint32_t)((gy + vrow0) * tiles_x + gx);
                    out_tile[o] = (KeyT)key;
                    out_gid[o] = gg;
                    if (hist) count_key(key, o);
                }
            }
            run += (uint32_t)__popcll(m);
        }
    }
    if (hist) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int e = q * kEmitThreads + tid, t = e >> 10, p = (e >> 8) & 3, d = e & 255;
            const uint32_t c = s_hist[t][p][d];
            if (c && x0 + (uint32_t)t < (uint32_t)kOsRanges) atomicAdd(&hz.ghist[((size_t)p * kOsRanges + x0 + t) * 256 + d], c);
        }
    }
}

// K6: per-tile [start, end) from the tile-sorted keys (ranges pre-zeroed)
// one 16-byte load (8 or 4 keys) + the key in front of it per thread
template <typename KeyT>
__global__ __launch_bounds__(256) void k_tile_ranges(uint32_t R, const KeyT* __restrict__ keys, uint2* __restrict__ ranges,
                                                     const unsigned long long* __restrict__ n_dev)
{
    if (n_dev) R = (uint32_t)min((unsigned long long)R, *n_dev);
    constexpr uint32_t KPV = 16 / (uint32_t)sizeof(KeyT);
    const uint32_t i0 = (blockIdx.x * 256u + threadIdx.x) * KPV;
    if (i0 >= R) return;
    uint32_t t[KPV];
    if (i0 + KPV <= R && (((uintptr_t)keys & 15) == 0)) {
        const uint4 q = *reinterpret_cast<const uint4*>(keys + i0);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (uint32_t c = 0; c < KPV; c++) t[c] = sizeof(KeyT) == 4 ? w[c] : ((w[c / 2] >> (16 * (c & 1))) & 0xffffu);
    } else {
#pragma unroll
        for (uint32_t c = 0; c < KPV; c++) t[c] = i0 + c < R ? (uint32_t)keys[i0 + c] : 0u;
    }
    uint32_t prev = i0 ? (uint32_t)keys[i0 - 1] : 0xffffffffu;
#pragma unroll
    for (uint32_t c = 0; c < KPV; c++) {
        const uint32_t i = i0 + c;
        if (i < R) {
            if (i == 0) ranges[t[c]].x = 0;
            else if (prev != t[c]) { ranges[prev].y = i; ranges[t[c]].x = i; }
            if (i == R - 1) ranges[t[c]].y = R;
            prev = t[c];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Direct binning (round 4): the tile lists WITHOUT an instance stream and without a tile sort.
// The instance list is the stable partition, by tile, of the (Gaussian, tile) pairs taken in depth order.  The sort route writes
// the pairs out as 6-byte records (k_emit), moves them twice (two 6-bit onesweep passes) and reads the keys once more for the
// ranges; here every pair is written exactly once, 4 bytes, at its final place:
//   k_chunk_counts   the depth-ordered Gaussians are cut into NC chunks of S (a multiple of 64), one WAVE per chunk: the gather of the
//                    tile records into depth order (as k_tile_counts) and the chunk's per-tile histogram, M[chunk][tile] (u16);
//   k_chunk_scan1/2  exclusive prefix of M down each tile's column (two levels: inside groups of Cg chunks, then over the groups, a block
//                    per 256-tile slab, which also forms the tile bases = the ranges; R for the host and the small clears ride along);
//   k_chunk_scatter  one wave per chunk again, with the chunk's row of start positions as an LDS counter table and a 64-bit lane
//                    mask per tile: per step of 64 Gaussians every lane ORs its bit into the masks of its tiles, then reads each mask
//                    back -- position = counter + popcount(mask below my lane) -- and the lowest lane of a mask advances the counter
//                    and clears the mask.  OR and add commute: the list is the sort's list bit for bit, with no atomics' order in it.
// LDS per scatter wave: 12 bytes per tile (26 kB at 980x545: six waves per CU, the whole chunk set resident at 1 M Gaussians); frames
// of more than kDbMaxTiles tiles (and 32-bit tile keys) keep the sort route.  Chunk c of the scatter runs on XCD (c / ceil(NC / 8)):
// an XCD owns a contiguous eighth of the depth order, so the 4-byte stores it scatters over a tile's segment fill whole lines of
// ITS L2 before they leave it.
// ------------------------------------------------------------------------------------------------
constexpr int kDbMaxTiles = 4096;
constexpr int kDbCountWaves = 4;      // chunks per workgroup of k_chunk_counts (kEmitThreads / 64: balance_build shares the launch)
struct DirectBin {
    int N, T, Tp /* T rounded up to 64 */, S, NC, G, Cg;
    int NS, Ts, Tsp, slab_rows;   // round 5: the tile grid cut into NS slabs of slab_rows tile rows (Ts tiles, Tsp = Ts rounded up to 64); 1 slab = the whole frame
    uint16_t* M;        // [NC][Tp]  per-chunk tile counts -> exclusive prefixes inside the chunk's group
    uint32_t* GT;       // [G][Tp]   group totals -> absolute start of the group inside the tile's segment
    uint32_t* tbase;    // [T + 1]   tile bases (saturated at 2^32 - 1)
    uint32_t* bsum;     // [G][slabs] instances per (group, 256-tile slab)
};

// ------------------------------------------------------------------------------------------------
// Lists cut where the tiles stopped last time (round 6).  At the headline the forward blends stage R_eff = 0.75 M of the R = 4.5 M
// instances the binning places (17 %; 4 % at 4 M Gaussians): a tile whose 256 pixels saturate after the nearest few hundred splats
// never reads the rest of its list.  The frame is recognised anyway (the view-cost cache of the balanced placement), so every tile
// remembers the DEPTH up to which its four waves staged instances at the frame's previous render, and the scatter -- the expensive
// half of the direct binning -- leaves out what lies behind:
//   * nothing about the LAYOUT changes: counts, scans, tile bases and R are the full ones, every pair keeps the position the full
//     binning gives it; lists are simply only WRITTEN up to a chunk boundary of the depth order, and the blends are handed the end
//     of the valid prefix as the tile's range end (from the chunk tables: no counting).
//   * one cut for the frame, exceptions for the few: tile t's own cut is the last chunk that starts at or in front of (the depth it
//     needed) x (1 + margin); C* is the chunk behind which at most `dmax` tiles' cuts lie (k_chunk_scan2: a histogram of the cuts in
//     LDS, every block for itself).  The chunks up to C* are scattered exactly as ever -- for EVERY tile, so all but the `dmax`
//     deepest tiles keep more than they asked for -- and a scatter wave of a chunk behind C* only serves the few "deep" tiles whose own
//     cut lies behind it (cut_chunk_tiles: the chunk's records against a handful of tiles, ~1 % of the scatter's work).  A tile that
//     reached the end of its full list with a live pixel (kCutOpenKey) is a deep tile that keeps everything.
//   * the speculation is verified ON THE DEVICE and repaired there: a blend wave that reaches the end of a cut list with a live
//     pixel flags its tile; two launches that follow every cut render -- the flagged tiles' left-out chunks (the same cut_chunk_tiles)
//     and the same blend over the flagged tiles' full lists -- find nothing to do (a word read per workgroup) unless a tile was
//     flagged.  The host never learns of it and never waits: image, radii, the lists' valid prefixes, checkpoints and gradients are
//     those of the full binning, bit for bit, by construction -- the instances a pixel blends are the same instances in the same
//     order (tests/test_gpu_listcut.py).
// What changes is what lies in the list buffer BEHIND a tile's valid prefix (stale words nobody reads) and ranges[t].y.
// Off ("list_cut" 0, a frame seen for the first time, another model under the same frame, the tile-sort / sort / slabbed / batched
// routes): every tile's cut is the last chunk and the code below is the round-5 code.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kCutOpenKey = 0xffffffffu;   // a tile that reached the end of its FULL list with a live pixel: never cut
struct ListCut {            // kernel-argument bundle; key == nullptr: off
    uint32_t* key;          // [kVcEntries][T] (cache): depth key up to which tile t's waves staged at the entry's last render
    uint32_t* owner_n;      // [kVcEntries] (cache): the model size those keys belong to (another model under the same frame: no cut)
    uint32_t* stats;        // [8] (cache): [0] cut renders, [1] renders that needed a repair, [2] tiles repaired, [3] chunks left to the deep tiles (sum), [4] deep tiles (sum)
    const uint32_t* cur;    // the placement's per-call words: [0] the frame's cache entry, [9] the entry was a hit (written by k_chunk_counts' builder 0)
    uint16_t* chunk;        // [Tp] per call: tile t's OWN cut: the last chunk of the depth order it asks for (what is written: max(that, C*))
    uint32_t* hist;         // [NC] per call: tiles per own cut
    uint32_t* tend;         // [Tp] per call: end of tile t's valid prefix (absolute list position)
    uint32_t* flag;         // [Tp] per call: tile t ran out of its cut list with a live pixel
    uint32_t* ctl;          // [8] per call: [0] C* (written by the cut pass for the repair pass), [1] deep tiles, [2] tiles flagged
    uint32_t* bkey;         // [NC] per call: depth key of the first Gaussian of every chunk
    const uint32_t* skey;   // the depth keys in depth order
    const uint32_t* tbase;  // [T + 1] the tile bases of the FULL binning (DirectBin::tbase)
    float margin;           // relative slack on the remembered depth
    int dmax;               // tiles that may keep more than the frame's cut
    int enable;             // 0: tables are written for "everything kept" (the caller asked for full lists on a route that can cut)
};

// all 64 lanes walk the candidate tiles of ONE large rect (more than 32 tiles: no mask in its TileRec), 64 at a time, in the
// emission's order (row-major over the tight rect, the exact test per tile): f(accepted, tile key) on every lane, every round
template <typename F>
__device__ __forceinline__ void big_rect_tiles(const Splat& s, uint32_t rect, int W, int H, int tiles_x, int tiles_y, int lane, F&& f)
{
    int x0, y0, x1, y1;
    tile_rect_tight(s.px, s.py, s.radius, s.ca, s.cb, s.cc, s.op, W, H, tiles_x, tiles_y, x0, y0, x1, y1);   // as k_preprocess
    const int vrow0 = (int)((rect >> 12) & 0xfffu) - y0;   // batched render: first tile row of this Gaussian's model (0 otherwise)
    const int ww = x1 - x0, full = ww * (y1 - y0);
    const TileTest tt = make_tile_test(s.px, s.py, s.ca, s.cb, s.cc, s.op);
    for (int c0 = 0; c0 < full; c0 += 64) {
        const int c = c0 + lane;
        bool ok = false;
        uint32_t key = 0u;
        if (c < full) {
            const int ty = c / ww, tx = c - ty * ww;
            ok = tile_accept(tt, x0 + tx, y0 + ty, W, H);
            key = (uint32_t)((y0 + ty + vrow0) * tiles_x + x0 + tx);
        }
        f(ok, key);
    }
}

// the tiles of a small rect, in mask order: f(tile key).  Two tiles per iteration: the walk of a lone wave is a chain of dependent
// instructions (lowest bit -> row -> key -> address), two independent chains per trip nearly halve it.
template <typename F>
__device__ __forceinline__ void small_rect_tiles(const TileRec& r, int tiles_x, F&& f)
{
    const uint32_t ww = (r.rect >> 24) & 63u, rx0 = r.rect & 0xfffu, ry0 = (r.rect >> 12) & 0xfffu;
    const uint32_t rcp = (65536u + ww - 1u) / (ww ? ww : 1u);   // floor(pos / ww) = pos * rcp >> 16 for pos < 32 <= 65536 / ww
    const uint32_t base = ry0 * (uint32_t)tiles_x + rx0, skip = (uint32_t)tiles_x - ww;   // key = base + pos + (pos / ww) * (tiles_x - ww)
    for (uint32_t m = r.mask; m != 0u;) {
        const uint32_t m1 = m & (m - 1u);
        const uint32_t p0 = (uint32_t)__builtin_ctz(m);
        f(base + p0 + ((p0 * rcp) >> 16) * skip);
        if (m1 != 0u) {
            const uint32_t p1 = (uint32_t)__builtin_ctz(m1);
            f(base + p1 + ((p1 * rcp) >> 16) * skip);
        }
        m = m1 & (m1 - 1u);
    }
}

// inclusive prefix sum over the 64 lanes by DPP row shifts and row broadcasts (no LDS round trips: ~8 instructions)
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    auto dpp = [](uint32_t v, auto ctrl, auto rmask, auto bmask) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, decltype(ctrl)::value, decltype(rmask)::value, decltype(bmask)::value, false);
    };
    using std::integral_constant;
    uint32_t r = x;
    r += dpp(x, integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xf>{});   // row_shr:1
    r += dpp(x, integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xf>{});   // row_shr:2
    r += dpp(x, integral_constant<int, 0x113>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xf>{});   // row_shr:3
    r += dpp(r, integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xe>{});   // row_shr:4, banks 1-3
    r += dpp(r, integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xc>{});   // row_shr:8, banks 2-3
    r += dpp(r, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{}, integral_constant<int, 0xf>{});   // row_bcast:15 -> rows 1, 3
    r += dpp(r, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{}, integral_constant<int, 0xf>{});   // row_bcast:31 -> rows 2, 3
    return r;
#else
    return x;
#endif
}

// sorted_gid == nullptr (the tile-sort route, round 5): the chunks cut the Gaussians in INDEX order -- no depth sort in front of the
// binning; the tile lists then come out in index order and k_tile_sort orders each by depth -- the records are read where they lie (no
// gather, no sorted copy) and the launch's last workgroup carries the rider that publishes R (radix_sort.h OsRider: on the other route
// it rides on the depth sort's first kernel).
__global__ __launch_bounds__(kEmitThreads) void k_chunk_counts(DirectBin db, int W, int H, int tiles_x, int tiles_y,
                                                               const uint32_t* __restrict__ sorted_gid, const TileRec* __restrict__ tilerec,
                                                               const Splat* __restrict__ splat, TileRec* __restrict__ sorted_rec, BlendBalance bb,
                                                               OsRider rider = OsRider{}, ListCut cut = ListCut{})
{
    const int nbuild = bb.hdr ? 8 : 0, c = (int)blockIdx.x - nbuild;
    if (c < 0) { balance_build(bb, (int)blockIdx.x); return; }
    if (cut.key && threadIdx.x == 0) {     // list cut: where every chunk of the depth order starts, and this call's control words
        cut.bkey[c] = cut.skey[(size_t)c * db.S];
        cut.hist[c] = 0u;
        if (c == 0) { cut.ctl[0] = (uint32_t)(db.NC - 1); cut.ctl[1] = 0u; cut.ctl[2] = 0u; }
    }
    if (rider.host && blockIdx.x == gridDim.x - 1) {   // (block-uniform)
        __shared__ unsigned long long s_r[2][16];
        onesweep_rider_publish(rider, s_r);
    }
    // one workgroup per chunk: its four waves take the chunk's steps of 64 Gaussians in turn and count into ONE table (adds commute)
    extern __shared__ uint32_t s_h[];           // [Tp / 2]: two u16 counters per word (a chunk holds fewer than 65 536 Gaussians)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < db.Tp / 2; i += kEmitThreads) s_h[i] = 0u;
    __syncthreads();
    const int j0 = c * db.S + wave * 64, j1 = min(db.N, c * db.S + db.S);
    constexpr int kStride = 64 * kDbCountWaves;
    // two loads ahead: the index of this wave's step after next and the record of its next step are in flight while a step is counted
    auto load_gid = [&](int j) -> uint32_t { return j < j1 ? (sorted_gid ? sorted_gid[j] : (uint32_t)j) : 0xffffffffu; };
    auto load_rec = [&](uint32_t g) -> TileRec { TileRec r; r.mask = 0u; r.rect = 1u << 24; if (g != 0xffffffffu) r = tilerec[g]; return r; };
    uint32_t gA = load_gid(j0 + lane), gB = load_gid(j0 + kStride + lane);
    TileRec rA = load_rec(gA);
    for (int j = j0 + lane; j - lane < j1; j += kStride) {
        const uint32_t g = gA;
        const TileRec r = rA;
        gA = gB;
        rA = load_rec(gA);
        gB = load_gid(j + 2 * kStride);
        if (sorted_rec && j < j1) sorted_rec[j] = r;
        const bool big = (r.rect & kTileRecBig) != 0u;
        if (!big) small_rect_tiles(r, tiles_x, [&](uint32_t t) { atomicAdd(&s_h[t >> 1], 1u << (16u * (t & 1u))); });
        for (unsigned long long bm = __ballot(big && r.mask != 0u); bm != 0ull; bm &= bm - 1ull) {
            const int b = (int)__builtin_ctzll(bm);
            const uint32_t gg = (uint32_t)__builtin_amdgcn_readlane((int)g, b), rect = (uint32_t)__builtin_amdgcn_readlane((int)r.rect, b);
            const Splat s = splat[gg];
            big_rect_tiles(s, rect, W, H, tiles_x, tiles_y, lane, [&](bool ok, uint32_t t) { if (ok) atomicAdd(&s_h[t >> 1], 1u << (16u * (t & 1u))); });
        }
    }
    __syncthreads();
    uint32_t* const row = reinterpret_cast<uint32_t*>(db.M + (size_t)c * db.Tp);
    for (int i = tid; i < db.Tp / 2; i += kEmitThreads) row[i] = s_h[i];
}

// level 1 (grid: 256-tile slabs x groups): thread (tile, group) turns its group's counts into exclusive prefixes, in place, and writes
// the group's total; the block's sum of those goes to bsum[group][slab]
__global__ __launch_bounds__(256) void k_chunk_scan1(DirectBin db)
{
    __shared__ uint32_t s_w[4];
    const int tid = threadIdx.x, t = (int)(blockIdx.x * 256u) + tid, grp = (int)blockIdx.y;
    uint32_t run = 0u;
    if (t < db.Tp) {
        const int c0 = grp * db.Cg, c1 = min(db.NC, c0 + db.Cg);
        uint16_t* p = db.M + (size_t)c0 * db.Tp + t;
        int c = c0;
        for (; c + 8 <= c1; c += 8, p += 8 * (size_t)db.Tp) {
            uint32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = p[(size_t)k * db.Tp];
#pragma unroll
            for (int k = 0; k < 8; k++) { p[(size_t)k * db.Tp] = (uint16_t)run; run += v[k]; }
        }
        for (; c < c1; c++, p += db.Tp) { const uint32_t v = *p; *p = (uint16_t)run; run += v; }
        db.GT[(size_t)grp * db.Tp + t] = run;
    }
    const uint32_t inc = wave_inclusive_sum(run);
    if ((tid & 63) == 63) s_w[tid >> 6] = inc;
    __syncthreads();
    if (tid == 0) db.bsum[(size_t)grp * gridDim.x + blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// level 2 (one block per 256-tile slab): per tile, the group totals -> starts relative to the tile's base (in place; every group of
// a tile in flight at once) and the tile's total; the slab's offset from the block sums of level 1, the tile bases by a block scan.
// Block 0: R to the host.  The small clears are spread over the grid.
constexpr int kDbMaxGroups = 64;
__global__ __launch_bounds__(256) void k_chunk_scan2(DirectBin db, unsigned long long* __restrict__ total_out, ZeroJobs zj,
                                                     unsigned long long* __restrict__ host_out, unsigned long long host_seq,
                                                     const unsigned int* __restrict__ window_overflow, unsigned long long* __restrict__ late_out = nullptr,
                                                     ListCut cut = ListCut{})
{
    extern __shared__ uint32_t s_bkey[];     // [NC] (list cut): the chunks' first depth keys
    __shared__ unsigned long long s_lo[4], s_all[4];
    if (cut.key)
        for (int i = threadIdx.x; i < db.NC; i += 256) s_bkey[i] = cut.bkey[i];
    __shared__ uint32_t s_w[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slab = (int)blockIdx.x, nslab = (int)gridDim.x, t = slab * 256 + tid;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        uint32_t* z = static_cast<uint32_t*>(zj.p[j]);
        for (uint32_t q = (uint32_t)(slab * 256 + tid); q < zj.words[j]; q += (uint32_t)nslab * 256u) z[q] = 0u;
    }
    // this slab's offset (the instances of the slabs below it) and R, from the level-1 block sums
    unsigned long long lo = 0ull, all = 0ull;
    for (int i = tid; i < db.G * nslab; i += 256) {
        const uint32_t v = db.bsum[i];
        all += v;
        if (i % nslab < slab) lo += v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { lo += __shfl_xor(lo, off, 64); all += __shfl_xor(all, off, 64); }
    if (lane == 0) { s_lo[wave] = lo; s_all[wave] = all; }
    // the column of every tile of the slab
    uint32_t tot = 0u;
    if (t < db.Tp) {
        uint32_t v[kDbMaxGroups];
#pragma unroll
        for (int g = 0; g < kDbMaxGroups; g++) v[g] = g < db.G ? db.GT[(size_t)g * db.Tp + t] : 0u;
#pragma unroll
        for (int g = 0; g < kDbMaxGroups; g++) {
            if (g < db.G) db.GT[(size_t)g * db.Tp + t] = tot;
            tot += v[g];
        }
    }
    if (t >= db.T) tot = 0u;
    const uint32_t inc = wave_inclusive_sum(tot);
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    lo = s_lo[0] + s_lo[1] + s_lo[2] + s_lo[3];
    all = s_all[0] + s_all[1] + s_all[2] + s_all[3];
    uint32_t add = 0u;
#pragma unroll
    for (int w = 0; w < 4; w++) add += w < wave ? s_w[w] : 0u;
    // (beyond 2^32 instances the host fails the call: positions only have to stay in bounds)
    const uint32_t base_t = (uint32_t)min(lo + add + inc - tot, 0xffffffffull);
    if (t < db.T) db.tbase[t] = base_t;
    if (cut.key) {
        // the tile's own cut: the last chunk that starts at or in front of (the depth its waves reached last time) x (1 + margin), into
        // chunk[t] and a histogram over the chunks; the frame's cut C* is read off that histogram by the scatter's waves (cut_frame)
        if (t < db.T) {
            const uint32_t e = cut.cur[0] < (uint32_t)kVcEntries ? cut.cur[0] : 0u;
            const bool use = cut.enable && cut.cur[9] != 0u && cut.owner_n[e] == (uint32_t)db.N;
            uint32_t ck = (uint32_t)(db.NC - 1);
            const uint32_t k = use ? cut.key[(size_t)e * db.T + t] : kCutOpenKey;
            if (k != kCutOpenKey && k != 0u) {
                const float d = __uint_as_float(k);
                const uint32_t kc = __float_as_uint(fmaf(d, cut.margin, d));
                int a = 0, b = db.NC;            // chunks [0, a) start at or in front of kc
                while (a < b) { const int m = (a + b) >> 1; if (s_bkey[m] <= kc) a = m + 1; else b = m; }
                ck = (uint32_t)max(a - 1, 0);
            }
            cut.chunk[t] = (uint16_t)ck;
            cut.flag[t] = 0u;
            if (use) atomicAdd(&cut.hist[ck], 1u);       // (nothing to cut: no histogram -- 2 000 adds to ONE word took 20 us)
            if (t == 0) cut.ctl[3] = use ? 1u : 0u;
        }
    }
    if (slab == 0 && tid == 0) {
        db.tbase[db.T] = (uint32_t)min(all, 0xffffffffull);
        *total_out = all;
        if (host_out) {   // as block_scan_body
            __atomic_store_n(host_out + 2, (unsigned long long)g_onesweep_giveups, __ATOMIC_RELAXED);
            __atomic_store_n(host_out + 3, (unsigned long long)(window_overflow ? *window_overflow : 0u), __ATOMIC_RELAXED);
            __atomic_store_n(host_out, all, __ATOMIC_RELAXED);
            __threadfence_system();
            __atomic_store_n(host_out + 1, host_seq, __ATOMIC_RELAXED);
            __threadfence_system();
        }
        if (late_out) scan_report_late(late_out, all, host_seq);
    }
}

// (Measured and dropped: both levels in ONE launch, the last block of a slab / of the grid to arrive -- a ticket behind a
//  __threadfence() -- doing the next level: 47 us against 5.5 + 14.5 (level 2 was one workgroup then).  An agent-scope release writes the XCD's whole L2 back, and the
//  chunk tables the level-1 blocks have just rewritten are 6 MB of dirty lines.)
// One wave per chunk.  LDS: a 64-bit lane mask and a position counter per tile, and the step's (tile, owner lane) pairs.
// Per step of 64 depth-consecutive Gaussians:
//   owners   every lane walks ITS record's tiles: ORs its bit into the tile's mask and appends (tile, lane) to the pair buffer at its
//            exclusive prefix -- no reads, nothing waits;
//   place    the pairs are dealt to the lanes 64 at a time (a lane's tile count does not matter any more: a wave of records with
//            4.5 tiles on average and one of 30 takes five rounds, not thirty): position = counter + popcount(mask below the owner);
//   advance  every pair adds one to its tile's counter and clears the mask (adds commute; nothing is read).
// A single wave's LDS operations execute in program order, so the three parts need no barrier between them -- only the compiler
// has to keep them apart.  Large rects (no mask in the record) are walked by all 64 lanes, one Gaussian at a time, in the same three
// parts.  The list is the sort route's list bit for bit.
constexpr int kDbPairs = 1024;        // pair buffer (a step whose small rects hold more is cut into runs of lanes that fit)
__device__ __forceinline__ void lds_order() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// The frame's cut C*: the chunk behind which at most `dmax` tiles' own cuts lie, read off the histogram k_chunk_scan2 built (one wave;
// every wave of the scatter does this for itself -- a few coalesced loads -- instead of a launch or a cross-block step in the scan).
// deep = the tiles whose own cut lies behind C*.
__device__ __forceinline__ uint32_t cut_frame(const ListCut& cut, const int NC, uint32_t& deep)
{
    const int lane = threadIdx.x & 63;
    deep = 0u;
    if (cut.ctl[3] == 0u) return (uint32_t)(NC - 1);      // nothing is cut this render (k_chunk_scan2 says so): no histogram was built
    // every word of the histogram requested before the first is used (the dependent form -- load, scan, decide, next 64 -- was 1.5 us
    // per round trip in front of every scatter wave); beyond 32 x 64 chunks the remainder goes the slow way
    constexpr int kR = 32;
    uint32_t v[kR];
#pragma unroll
    for (int q = 0; q < kR; q++) { const int c = NC - 1 - 64 * q - lane; v[q] = (64 * q < NC && c >= 0) ? cut.hist[c] : 0u; }
    uint32_t acc = 0u;
#pragma unroll
    for (int q = 0; q < kR; q++) {
        if (64 * q >= NC) break;
        const int c = NC - 1 - 64 * q - lane;
        const uint32_t inc = acc + wave_inclusive_sum(v[q]);
        const unsigned long long over = __ballot(c >= 0 && inc > (uint32_t)cut.dmax);
        if (over) {
            const int l = (int)__builtin_ctzll(over);
            deep = (uint32_t)__builtin_amdgcn_readlane((int)(inc - v[q]), l);
            return (uint32_t)(NC - 1 - 64 * q - l);
        }
        acc = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
    for (int top = NC - 1 - 64 * kR; top >= 0; top -= 64) {
        const int c = top - lane;
        const uint32_t w = c >= 0 ? cut.hist[c] : 0u;
        const uint32_t inc = acc + wave_inclusive_sum(w);
        const unsigned long long over = __ballot(c >= 0 && inc > (uint32_t)cut.dmax);
        if (over) {
            const int l = (int)__builtin_ctzll(over);
            deep = (uint32_t)__builtin_amdgcn_readlane((int)(inc - w), l);
            return (uint32_t)(top - l);
        }
        acc = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
    deep = acc;
    return 0u;
}

// The pairs of a FEW tiles in one chunk of the depth order, written where the full binning puts them: position = the tile's base + the
// chunk's start inside the tile (the chunk tables) + the number of the chunk's earlier Gaussians that touch the tile.  One wave; the
// chunk's records 64 at a time, lane = Gaussian (two steps of loads in flight), against every wanted tile in turn (a rect test and a
// mask bit; a large rect -- no mask in its record -- evaluates the exact tile test on that one tile).  want(t) says which tiles: the deep
// tiles of a cut render whose own cut lies at or behind this chunk, or the flagged tiles of a repair pass whose cut lies in front of it.
template <typename Want>
__device__ __forceinline__ void cut_chunk_tiles(const DirectBin& db, const int c, const int W, const int H, const int tiles_x, const int tiles_y,
                                                const uint32_t* __restrict__ sorted_gid, const TileRec* __restrict__ sorted_rec,
                                                const Splat* __restrict__ splat, uint32_t* __restrict__ list, const uint32_t cap,
                                                uint32_t* s_txy /*[64]*/, uint32_t* s_pos /*[64]*/, Want&& want)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = lanemask_lt();
    const int j0 = c * db.S, j1 = min(db.N, j0 + db.S);
    const uint32_t* gt = db.GT + (size_t)(c / db.Cg) * db.Tp;
    const uint16_t* mr = db.M + (size_t)c * db.Tp;
    auto load_rec = [&](int j) -> TileRec { TileRec r; r.mask = 0u; r.rect = 1u << 24; if (j < j1) r = sorted_rec[j]; return r; };
    auto load_gid = [&](int j) -> uint32_t { return j < j1 ? sorted_gid[j] : 0u; };
    // which tiles are wanted: asked for ALL tiles at once (bit q of `mine` = tile 64 q + lane), every load in flight together -- asked
    // block by block, with a ballot and a branch between the loads, this was 34 memory round trips in a row in front of every chunk
    unsigned long long mine = 0ull;
#pragma unroll
    for (int q = 0; q < kDbMaxTiles / 64; q++) {
        const int t = 64 * q + lane;
        if (64 * q < db.T && t < db.T && want((uint32_t)t)) mine |= 1ull << q;
    }
    for (int t0 = 0; t0 < db.T;) {
        // the next (up to) 64 wanted tiles, in tile order: whole blocks of 64 tiles as long as they fit
        int nt = 0;
        while (t0 < db.T) {
            const int t = t0 + lane;
            const bool w = ((mine >> (t0 >> 6)) & 1ull) != 0ull;
            const unsigned long long b = __ballot(w);
            const int got = (int)__popcll(b);
            if (nt + got > 64) break;            // (nt > 0 here: a block holds at most 64) -- the next round starts with this block
            if (w) {
                const int k = nt + (int)__popcll(b & lt);
                s_txy[k] = ((uint32_t)(t / tiles_x) << 16) | (uint32_t)(t % tiles_x);
                s_pos[k] = db.tbase[t] + gt[t] + (uint32_t)mr[t];
            }
            nt += got;
            t0 += 64;
        }
        if (nt == 0) break;
        lds_order();
        TileRec rA = load_rec(j0 + lane), rB = load_rec(j0 + 64 + lane);
        uint32_t gA = load_gid(j0 + lane), gB = load_gid(j0 + 64 + lane);
        for (int j = j0; j < j1; j += 64) {
            const TileRec r = rA;
            const uint32_t g = gA;
            rA = rB; gA = gB;
            rB = load_rec(j + 128 + lane);
            gB = load_gid(j + 128 + lane);
            const bool big = (r.rect & kTileRecBig) != 0u && r.mask != 0u;
            const uint32_t ww = (r.rect >> 24) & 63u, rx0 = r.rect & 0xfffu, ry0 = (r.rect >> 12) & 0xfffu;
            TileTest tt = {};
            int bx0 = 0, by0 = 0, bx1 = 0, by1 = 0;
            if (big) {
                const Splat sp = splat[g];
                tile_rect_tight(sp.px, sp.py, sp.radius, sp.ca, sp.cb, sp.cc, sp.op, W, H, tiles_x, tiles_y, bx0, by0, bx1, by1);
                tt = make_tile_test(sp.px, sp.py, sp.ca, sp.cb, sp.cc, sp.op);
            }
            const bool any_big = __ballot(big) != 0ull;
            for (int k = 0; k < nt; k++) {
                const uint32_t txy = s_txy[k], tx = txy & 0xffffu, ty = txy >> 16;
                const uint32_t dx = tx - rx0, dy = ty - ry0, bit = dy * ww + dx;      // (unsigned: a tile left of / above the rect wraps)
                bool in = !big && dx < ww && bit < 32u && ((r.mask >> bit) & 1u) != 0u;
                if (any_big && big) in = (int)tx >= bx0 && (int)tx < bx1 && (int)ty >= by0 && (int)ty < by1 && tile_accept(tt, (int)tx, (int)ty, W, H);
                const unsigned long long b = __ballot(in);
                if (b == 0ull) continue;
                const uint32_t p0 = s_pos[k];
                if (in) { const uint32_t pos = p0 + (uint32_t)__popcll(b & lt); if (pos < cap) list[pos] = g; }
                lds_order();
                if (lane == 0) s_pos[k] = p0 + (uint32_t)__popcll(b);
                lds_order();
            }
        }
        lds_order();
    }
}

#ifdef GSR_DB_TIMING
__device__ unsigned long long g_db_dbg[16];   // s_memtime ticks per part, summed over the waves of every launch (tools/db_timing.sh)
#define DB_T(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); if (lane == 0) dbt[k] += now_ - dbt_last; dbt_last = now_; } while (0)
#else
#define DB_T(k) do { } while (0)
#endif
// SLAB (round 5, frames above kDbMaxTiles tiles): the tile grid is cut into db.NS slabs of whole tile rows and a chunk is walked by NS
// waves, each with the tables of ITS slab in LDS and blind to every other tile: a small rect's mask is cut down to the slab's rows when
// the record is loaded (whole tile rows: a contiguous bit range), a large rect is always walked outside the pair buffer (its count
// inside a slab is not known without the walk) with its tiles tested against the slab.  Every tile belongs to one slab and the
// chunks of a slab are the chunks of the frame, so the list is the same list.
// PAIRS (the tile-sort route, round 5): the chunks cut the Gaussians in index order (sorted_gid == nullptr, sorted_rec = the records where
// the preprocess left them) and what is placed is the PAIR (depth key, Gaussian) -- eight bytes per store instead of four -- into
// `pairs`; k_tile_sort orders every tile's pairs by key and writes the list.
// CUT (round 6, see ListCut): 1 = the cut pass -- the chunks up to C* are scattered as ever, a wave of a chunk behind C* serves the few deep
// tiles whose own cut lies at or behind its chunk (cut_chunk_tiles), and the ranges end at the tiles' valid prefixes; 2 = the repair
// pass: nothing unless a blend wave flagged a tile, then the flagged tiles' pairs in the chunks behind their cuts.
template <bool SLAB, bool PAIRS = false, int CUT = 0>
__global__ __launch_bounds__(64) void k_chunk_scatter(DirectBin db, int W, int H, int tiles_x, int tiles_y,
                                                      const uint32_t* __restrict__ sorted_gid, const TileRec* __restrict__ sorted_rec,
                                                      const Splat* __restrict__ splat, uint32_t* __restrict__ list, uint2* __restrict__ ranges,
                                                      uint32_t cap, const uint32_t* __restrict__ dkey = nullptr, uint2* __restrict__ pairs = nullptr,
                                                      ListCut cut = ListCut{})
{
    static_assert(!CUT || (!SLAB && !PAIRS), "the list cut serves the plain direct binning");
    extern __shared__ unsigned long long s_dyn[];
    const int LT = SLAB ? db.Tsp : db.Tp;                                      // tiles this wave keeps tables for
    unsigned long long* const s_mask = s_dyn;                                  // [LT]
    uint32_t* const s_cnt = reinterpret_cast<uint32_t*>(s_dyn + LT);          // [LT]
    uint32_t* const s_pair = s_cnt + LT;                                       // [kDbPairs]
    const int lane = threadIdx.x, b = (int)blockIdx.x;
    if (CUT == 2 && cut.ctl[2] == 0u) return;     // the repair pass: no tile was flagged -- every wave of the launch leaves here
    uint32_t cstar = 0u, ndeep = 0u;
    if (CUT == 1) cstar = cut_frame(cut, db.NC, ndeep);
    if (CUT == 2) cstar = cut.ctl[0];
    {   // the ranges: tile bases clipped to the list's capacity (an overflowing speculative launch is run again)
        const int t = b * 64 + lane;
        if (t < db.T) {
            uint32_t tend = db.tbase[t + 1];
            if (CUT == 1) {    // what is written of tile t: the chunks up to its own cut or the frame's, whichever lies behind
                const uint32_t c1 = max((uint32_t)cut.chunk[t], cstar) + 1u;
                if (c1 < (uint32_t)db.NC) tend = db.tbase[t] + db.GT[(size_t)(c1 / (uint32_t)db.Cg) * db.Tp + t] + (uint32_t)db.M[(size_t)c1 * db.Tp + t];
                cut.tend[t] = tend;
            }
            const uint32_t lo = min(db.tbase[t], cap), hi = min(tend, cap);
            if (CUT != 2 || cut.flag[t]) ranges[t] = hi > lo ? make_uint2(lo, hi) : make_uint2(0u, 0u);   // (an empty tile reads (0, 0), as on the sort route)
            if (CUT == 1) {    // this render's blends rebuild the tile's key with atomic maxima (blend_fwd_item)
                const uint32_t e = cut.cur[0] < (uint32_t)kVcEntries ? cut.cur[0] : 0u;
                cut.key[(size_t)e * db.T + t] = 0u;
            }
        }
        if (CUT && b == 0 && lane == 0) {     // the books of the cut
            if (CUT == 1) {
                const uint32_t e = cut.cur[0] < (uint32_t)kVcEntries ? cut.cur[0] : 0u;
                cut.owner_n[e] = (uint32_t)db.N;
                cut.ctl[0] = cstar; cut.ctl[1] = ndeep;
                cut.stats[0] += 1u;
                cut.stats[3] += (uint32_t)(db.NC - 1) - min(cstar, (uint32_t)(db.NC - 1));
                cut.stats[4] += ndeep;
            } else { cut.stats[1] += 1u; cut.stats[2] += cut.ctl[2]; }
        }
    }
    // XCD x = b & 7 owns the chunks [x per, (x + 1) per); a chunk's NS slab waves sit next to each other (they read the same records)
    const int per = (db.NC + 7) >> 3, kk = b >> 3, c = (b & 7) * per + (SLAB ? kk / db.NS : kk), slab = SLAB ? kk % db.NS : 0;
    if ((SLAB ? kk / db.NS : kk) >= per || c >= db.NC) return;
    if (CUT && (uint32_t)c > cstar) {             // behind the frame's cut: only a few tiles want this chunk's pairs
        if (CUT == 1 && ndeep == 0u) return;                               // (no deep tile at all)
        uint32_t* const s_tile = s_pair;                                   // (the pair buffer is not in use on this path)
        uint32_t* const s_pos = s_pair + 64;
        if (CUT == 1) cut_chunk_tiles(db, c, W, H, tiles_x, tiles_y, sorted_gid, sorted_rec, splat, list, cap, s_tile, s_pos,
                                      [&](uint32_t t) { return (uint32_t)cut.chunk[t] >= (uint32_t)c; });
        else cut_chunk_tiles(db, c, W, H, tiles_x, tiles_y, sorted_gid, sorted_rec, splat, list, cap, s_tile, s_pos,
                             [&](uint32_t t) { return cut.flag[t] != 0u && (uint32_t)cut.chunk[t] < (uint32_t)c; });
        return;
    }
    if (CUT == 2) return;                         // (the chunks up to C* were written for every tile)
    const uint32_t t0 = SLAB ? (uint32_t)(slab * db.Ts) : 0u;                                   // first tile of the slab
    const uint32_t tn = SLAB ? (uint32_t)min(db.Ts, db.T - slab * db.Ts) : (uint32_t)db.T;       // tiles in it
    const int srow0 = slab * db.slab_rows, srow1 = srow0 + db.slab_rows;                         // its tile rows (SLAB)
#ifdef GSR_DB_TIMING
    unsigned long long dbt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dbt_last = __builtin_readcyclecounter();
#endif
    {
        const uint32_t* gt = db.GT + (size_t)(c / db.Cg) * db.Tp;
        const uint16_t* mr = db.M + (size_t)c * db.Tp;
        // (every row of a 980x545 frame's tables in flight at once -- the chunk's rows were written by other XCDs and come from memory,
        //  ~1.7 us a round trip: one row at a time this loop was a fifth of the kernel, sixteen at a time still three round trips)
        constexpr int kRows = 36;
        for (int i0 = 0; i0 < LT; i0 += 64 * kRows) {
            uint32_t tb[kRows], gb[kRows], mc[kRows];   // tile base, the group's start in the tile, the chunk's start in the group
#pragma unroll
            for (int q = 0; q < kRows; q++) {
                const int i = i0 + 64 * q + lane;       // tile inside the slab; t0 + i in the frame
                const bool v = i < LT && (uint32_t)i < tn + (SLAB ? 0u : (uint32_t)(db.Tp - db.T));
                const int g = (int)t0 + i;
                tb[q] = v && g < db.T ? db.tbase[g] : 0u;
                gb[q] = v ? gt[g] : 0u;
                mc[q] = v ? (uint32_t)mr[g] : 0u;
            }
#pragma unroll
            for (int q = 0; q < kRows; q++) {
                const int i = i0 + 64 * q + lane;
                if (i < LT) { s_cnt[i] = tb[q] + gb[q] + mc[q]; s_mask[i] = 0ull; }
            }
        }
    }
    lds_order();
    DB_T(0);
    const unsigned long long me = 1ull << lane, lt = lanemask_lt();
    const int j0 = c * db.S, j1 = min(db.N, j0 + db.S);
    auto load_gid = [&](int j) -> uint32_t { return j < j1 ? (PAIRS ? (uint32_t)j : sorted_gid[j]) : 0u; };
    auto load_rec = [&](int j) -> TileRec { TileRec r; r.mask = 0u; r.rect = 1u << 24; if (j < j1) r = sorted_rec[j]; return r; };
    auto load_key = [&](int j) -> uint32_t { return (PAIRS && j < j1) ? dkey[j] : 0u; };
    // two steps of loads in flight (a sorted record arrives from HBM after ~1.5 us under this kernel's traffic)
    uint32_t gA = load_gid(j0 + lane), gB = load_gid(j0 + 64 + lane);
    TileRec rA = load_rec(j0 + lane), rB = load_rec(j0 + 64 + lane);
    uint32_t kA = load_key(j0 + lane), kB = load_key(j0 + 64 + lane);
    for (int j = j0 + lane; j - lane < j1; j += 64) {
        const uint32_t g = gA;
        TileRec r = rA;
        const uint32_t kd = kA;   // (PAIRS) this lane's Gaussian's depth key
        gA = gB; rA = rB; kA = kB;
        gB = load_gid(j + 128);
        rB = load_rec(j + 128);
        kB = load_key(j + 128);
        if (SLAB && !(r.rect & kTileRecBig)) {   // the rect's rows inside the slab are a contiguous run of mask bits
            const int ww = (int)((r.rect >> 24) & 63u), ry0 = (int)((r.rect >> 12) & 0xfffu);
            const int lo = min(32, max(0, (srow0 - ry0) * ww)), hi = min(32, max(0, (srow1 - ry0) * ww));
            const uint32_t below_hi = hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u), below_lo = lo >= 32 ? 0xffffffffu : ((1u << lo) - 1u);
            r.mask &= below_hi & ~below_lo;
        }
        // A large rect (no mask in its record) is walked by all 64 lanes, once, and its accepted tiles go into the pair buffer like
        // everyone's; one of more tiles than the buffer holds ("huge") is walked three times instead, outside the buffer.
        const bool big = (r.rect & kTileRecBig) != 0u && r.mask != 0u, huge = big && (SLAB || r.mask > (uint32_t)kDbPairs);
        const unsigned long long bigs = __ballot(big && !huge), huges = __ballot(huge);
        const uint32_t cnt = huge ? 0u : tilerec_count(r);
        const uint32_t incl = wave_inclusive_sum(cnt);
        // runs of lanes whose pairs fit the buffer: ONE run unless the step holds more than kDbPairs pairs
        auto run_end = [&](int lo, uint32_t before) -> int {
            const unsigned long long over = __ballot(incl - before > (uint32_t)kDbPairs) & ~((1ull << lo) - 1ull);
            return over ? (int)__builtin_ctzll(over) : 64;
        };
        // the pairs of lanes lo .. hi-1 into the buffer (with_or: and their bits into the masks -- every lane's, whatever its run)
        auto owners = [&](int lo, int hi, uint32_t before, bool with_or) {
            const bool mine = lane >= lo && lane < hi;
            if (!big && (with_or || mine)) {
                uint32_t* o = s_pair + (incl - cnt - before);
                small_rect_tiles(r, tiles_x, [&](uint32_t tg) {
                    const uint32_t t = tg - t0;          // (index inside the slab; the mask was cut down to it)
                    if (with_or) atomicOr(&s_mask[t], me);
                    if (mine) *o++ = (t << 6) | (uint32_t)lane;
                });
            }
            for (unsigned long long bm = bigs; bm != 0ull; bm &= bm - 1ull) {
                const int bl = (int)__builtin_ctzll(bm);
                const bool in_run = bl >= lo && bl < hi;
                if (!with_or && !in_run) continue;
                const uint32_t gg = (uint32_t)__builtin_amdgcn_readlane((int)g, bl), rect = (uint32_t)__builtin_amdgcn_readlane((int)r.rect, bl);
                uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)(incl - cnt), bl) - before;
                const Splat s = splat[gg];
                big_rect_tiles(s, rect, W, H, tiles_x, tiles_y, lane, [&](bool ok, uint32_t tg) {
                    const uint32_t t = tg - t0;          // (this branch only runs without slabs: t0 = 0)
                    const unsigned long long acc = __ballot(ok);
                    if (ok) {
                        if (with_or) atomicOr(&s_mask[t], 1ull << bl);
                        if (in_run) s_pair[(o + (uint32_t)__popcll(acc & lt)) & (uint32_t)(kDbPairs - 1)] = (t << 6) | (uint32_t)bl;
                    }
                    o += (uint32_t)__popcll(acc);
                });
            }
        };
        int lo = 0, hi = run_end(0, 0u);
        uint32_t before = 0u;
        const bool one_run = hi == 64;
        DB_T(1);
        owners(0, hi, 0u, true);
        DB_T(2);
        for (unsigned long long bm = huges; bm != 0ull; bm &= bm - 1ull) {
            const int bl = (int)__builtin_ctzll(bm);
            const uint32_t gg = (uint32_t)__builtin_amdgcn_readlane((int)g, bl), rect = (uint32_t)__builtin_amdgcn_readlane((int)r.rect, bl);
            const Splat s = splat[gg];
            big_rect_tiles(s, rect, W, H, tiles_x, tiles_y, lane, [&](bool ok, uint32_t tg) { const uint32_t t = tg - t0; if (ok && t < tn) atomicOr(&s_mask[t], 1ull << bl); });
        }
        // the loads of the step after next have had a step and this owner loop to arrive; taken HERE, in front of this step's
        // scattered stores (a wait for a load is a wait for every store issued before it: vmcnt counts both)
        asm volatile("" : "+v"(gA), "+v"(rA.mask), "+v"(rA.rect));   // (the NEXT step's: issued a step ago)
        if (PAIRS) asm volatile("" : "+v"(kA));
        lds_order();
        DB_T(3);
        // place: the pairs, 64 at a time, four rounds per batch of LDS round trips
        for (;;) {
            const uint32_t P = (uint32_t)__builtin_amdgcn_readlane((int)incl, hi - 1) - before;
            // (straight-line code per batch size: with uniform branches between the rounds the compiler drains the LDS queue at every
            //  join, and the batch became one round trip per READ -- three thousand cycles per step, a third of the kernel)
            for (uint32_t p0 = 0u; p0 < P; p0 += 512u) {
                const uint32_t left = P - p0;
                auto batch = [&](auto rounds_tag) {
                    constexpr int RN = decltype(rounds_tag)::value;
                    uint32_t e[RN], go[RN], base[RN], ko[PAIRS ? RN : 1];
                    unsigned long long mk[RN];
#pragma unroll
                    for (int q = 0; q < RN; q++) e[q] = s_pair[(p0 + 64u * q + (uint32_t)lane) & (uint32_t)(kDbPairs - 1)];   // (in bounds whatever P is)
#pragma unroll
                    for (int q = 0; q < RN; q++) {
                        const bool v = 64u * q + (uint32_t)lane < left;
                        e[q] = v ? e[q] : 0xffffffffu;
                        const uint32_t t = v ? e[q] >> 6 : 0u;
                        go[q] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((e[q] & 63u) << 2), (int)g);   // (every lane: an owner's index is fetched from ITS lane)
                        if (PAIRS) ko[q] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((e[q] & 63u) << 2), (int)kd);
                        mk[q] = s_mask[t];
                        base[q] = s_cnt[t];
                    }
#pragma unroll
                    for (int q = 0; q < RN; q++)
                        if (e[q] != 0xffffffffu) {
                            const uint32_t pos = base[q] + (uint32_t)__popcll(mk[q] & ((1ull << (e[q] & 63u)) - 1ull));
                            if (pos < cap) {
                                if (PAIRS) pairs[pos] = make_uint2(ko[q], go[q]);
                                else list[pos] = go[q];   // (plain stores: merged in this XCD's L2; nontemporal ones measured 166 us against 50)
                            }
                        }
                };
                if (left <= 128u) batch(std::integral_constant<int, 2>{});
                else if (left <= 256u) batch(std::integral_constant<int, 4>{});
                else if (left <= 384u) batch(std::integral_constant<int, 6>{});
                else batch(std::integral_constant<int, 8>{});
            }
            if (hi == 64) break;
            lds_order();       // (rare) the next run of lanes refills the pair buffer
            before += P;
            lo = hi;
            hi = run_end(lo, before);
            owners(lo, hi, before, false);
            lds_order();
        }
        for (unsigned long long bm = huges; bm != 0ull; bm &= bm - 1ull) {
            const int bl = (int)__builtin_ctzll(bm);
            const uint32_t gg = (uint32_t)__builtin_amdgcn_readlane((int)g, bl), rect = (uint32_t)__builtin_amdgcn_readlane((int)r.rect, bl);
            const Splat s = splat[gg];
            big_rect_tiles(s, rect, W, H, tiles_x, tiles_y, lane, [&](bool ok, uint32_t tg) {
                const uint32_t t = tg - t0;
                if (ok && t < tn) {
                    const uint32_t pos = s_cnt[t] + (uint32_t)__popcll(s_mask[t] & ((1ull << bl) - 1ull));
                    if (pos < cap) {
                        if (PAIRS) pairs[pos] = make_uint2((uint32_t)__builtin_amdgcn_readlane((int)kd, bl), gg);
                        else list[pos] = gg;
                    }
                }
            });
        }
        lds_order();
        DB_T(4);
        // advance: every (Gaussian, tile) adds one to the tile's counter and clears the mask
        if (one_run) {   // (the pair buffer still holds the whole step: dealt to the lanes like the placement, nothing read back but the pairs)
            const uint32_t P = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            for (uint32_t p0 = 0u; p0 < P; p0 += 512u) {
                const uint32_t left = P - p0;
                auto batch = [&](auto rounds_tag) {
                    constexpr int RN = decltype(rounds_tag)::value;
                    uint32_t e[RN];
#pragma unroll
                    for (int q = 0; q < RN; q++) e[q] = s_pair[(p0 + 64u * q + (uint32_t)lane) & (uint32_t)(kDbPairs - 1)];
#pragma unroll
                    for (int q = 0; q < RN; q++)
                        if (64u * q + (uint32_t)lane < left) { atomicAdd(&s_cnt[e[q] >> 6], 1u); s_mask[e[q] >> 6] = 0ull; }
                };
                if (left <= 128u) batch(std::integral_constant<int, 2>{});
                else if (left <= 256u) batch(std::integral_constant<int, 4>{});
                else if (left <= 384u) batch(std::integral_constant<int, 6>{});
                else batch(std::integral_constant<int, 8>{});
            }
        } else {
            if (!big) small_rect_tiles(r, tiles_x, [&](uint32_t tg) { const uint32_t t = tg - t0; atomicAdd(&s_cnt[t], 1u); s_mask[t] = 0ull; });
            for (unsigned long long bm = bigs; bm != 0ull; bm &= bm - 1ull) {
                const int bl = (int)__builtin_ctzll(bm);
                const uint32_t gg = (uint32_t)__builtin_amdgcn_readlane((int)g, bl), rect = (uint32_t)__builtin_amdgcn_readlane((int)r.rect, bl);
                const Splat s = splat[gg];
                big_rect_tiles(s, rect, W, H, tiles_x, tiles_y, lane, [&](bool ok, uint32_t tg) { const uint32_t t = tg - t0; if (ok && t < tn) { atomicAdd(&s_cnt[t], 1u); s_mask[t] = 0ull; } });
            }
        }
        for (unsigned long long bm = huges; bm != 0ull; bm &= bm - 1ull) {
            const int bl = (int)__builtin_ctzll(bm);
            const uint32_t gg = (uint32_t)__builtin_amdgcn_readlane((int)g, bl), rect = (uint32_t)__builtin_amdgcn_readlane((int)r.rect, bl);
            const Splat s = splat[gg];
            big_rect_tiles(s, rect, W, H, tiles_x, tiles_y, lane, [&](bool ok, uint32_t tg) { const uint32_t t = tg - t0; if (ok && t < tn) { atomicAdd(&s_cnt[t], 1u); s_mask[t] = 0ull; } });
        }
        lds_order();
        DB_T(5);
    }
#ifdef GSR_DB_TIMING
    if (lane == 0) {
        for (int k = 0; k < 6; k++) atomicAdd(&g_db_dbg[k], dbt[k]);
        atomicAdd(&g_db_dbg[6], 1ull);
        atomicAdd(&g_db_dbg[7], (unsigned long long)((j1 - j0 + 63) / 64));
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// Tile-sort route (round 5): NO global depth sort.  The direct binning runs over the Gaussians in index order (k_chunk_counts /
// k_chunk_scatter<.., PAIRS>), every tile's (depth key, Gaussian) pairs land in its segment in index order, and ONE workgroup per tile
// sorts its segment by key -- stable, so equal keys stay in index order: exactly the order the stable global sort of (key, index) gave
// the tile, the list is the other routes' list bit for bit.  What it replaces at 1 M Gaussians: a histogram launch and three
// look-back passes over 1 M keys (71 us, latency-bound: a chain over 245 tiles per pass) and the 8-byte gather of the tile records
// in depth order; what it costs: 4.5 M pairs sorted where they are, in LDS, by 2 170 independent workgroups.
// Segments of up to 4 096 pairs are sorted from registers through one LDS buffer (the onesweep passes' ranking, radix_sort.h
// wave_rank: 8-bit digits, a digit all keys of the tile share is skipped); longer ones go through global memory a chunk at a time
// (pairs <-> pairs_alt), same passes.
// ------------------------------------------------------------------------------------------------
// Two launches share the tiles by length: single waves take the segments of up to 1 024 pairs (k_tile_sort_wave below), 1 024-thread
// workgroups the longer ones (up to 4 096 pairs from registers, beyond that a chunk of 4 096 at a time through global memory).  The
// second launch is a handful of persistent workgroups that walk the tile ranges for long segments (a frame has few or none: 17 360
// workgroups of 1 024 threads that leave at once took 18 us to dispatch).
constexpr int kTsCap = 4096, kTsSmallMax = 1024;

// GATHER (the route behind the emit path: batched renders, frames above 4 096 tiles): the segment holds Gaussian indices only, in index
// order (the stable tile-key sort kept it); the keys are fetched from the per-Gaussian depth keys.  `dst` is then the segment itself:
// every index is read before the first barrier, every result written behind the last.
template <int THREADS, int IPT, bool GATHER = false>
__device__ __forceinline__ void tile_sort_regs(const uint2* __restrict__ src, uint32_t* dst, uint32_t n, const uint32_t* __restrict__ dkey,
                                               unsigned long long (*s_mask)[256], uint32_t (*s_cnt)[256], uint32_t* s_keys, uint32_t* s_vals,
                                               uint32_t* s_start, uint32_t* s_wsum, uint32_t* s_bits)
{
    constexpr int WAVES = THREADS / 64;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    uint32_t key[IPT], val[IPT], dig[IPT], rnk[IPT];
    uint32_t k_or = 0u, k_and = 0xffffffffu;
#pragma unroll
    for (int r = 0; r < IPT; r++) {
        const uint32_t p = (uint32_t)(wave * (IPT * 64) + r * 64 + lane);   // wave-major: the order wave_rank keeps
        if (GATHER) val[r] = p < n ? dst[p] : 0u;
        else if (p < n) { const uint2 e = src[p]; key[r] = e.x; val[r] = e.y; }
        if (!GATHER && p >= n) { key[r] = 0xffffffffu; val[r] = 0u; }   // padding: behind every real pair (a key is the bit pattern of a positive float), and it stays there
    }
    if (GATHER) {
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const uint32_t p = (uint32_t)(wave * (IPT * 64) + r * 64 + lane);
            key[r] = p < n ? dkey[val[r]] : 0xffffffffu;
        }
    }
#pragma unroll
    for (int r = 0; r < IPT; r++) {
        const uint32_t p = (uint32_t)(wave * (IPT * 64) + r * 64 + lane);
        if (p < n) { k_or |= key[r]; k_and &= key[r]; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { k_or |= (uint32_t)__shfl_xor((int)k_or, off, 64); k_and &= (uint32_t)__shfl_xor((int)k_and, off, 64); }
    if (lane == 0) { s_bits[wave] = k_or; s_bits[WAVES + wave] = k_and; }
    __syncthreads();
    uint32_t differ = 0u;
    {
        uint32_t o = 0u, a = 0xffffffffu;
#pragma unroll
        for (int w = 0; w < WAVES; w++) { o |= s_bits[w]; a &= s_bits[WAVES + w]; }
        differ = o & ~a;   // bits in which two of the tile's keys differ
    }
#pragma unroll 1
    for (int shift = 0; shift < 32; shift += 8) {
        if (((differ >> shift) & 0xffu) == 0u) continue;   // (uniform) every key of the tile has the same digit here
#pragma unroll
        for (int r = 0; r < IPT; r++) dig[r] = (key[r] >> shift) & 0xffu;
        wave_rank<IPT, 256>(s_mask[wave], s_cnt[wave], dig, rnk, lane);
        __syncthreads();
        uint32_t mine = 0u;   // threads 0..255 = digits: the waves' counts -> exclusive prefixes in wave order, and the digit's total
        if (tid < 256) {
#pragma unroll
            for (int w = 0; w < WAVES; w++) { const uint32_t c = s_cnt[w][tid]; s_cnt[w][tid] = mine; mine += c; }
        }
        const uint32_t ex = block_scan_excl<WAVES>(mine, s_wsum, tid);   // (threads beyond the digits add zero behind them)
        if (tid < 256) s_start[tid] = ex;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const uint32_t lp = s_start[dig[r]] + s_cnt[wave][dig[r]] + rnk[r];
            s_keys[lp] = key[r]; s_vals[lp] = val[r];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const uint32_t p = (uint32_t)(wave * (IPT * 64) + r * 64 + lane);
            key[r] = s_keys[p]; val[r] = s_vals[p];
        }
        __syncthreads();   // (the key / value buffer IS the ranking's mask tables: nobody clears a table while a neighbour still reads pairs)
    }
#pragma unroll
    for (int r = 0; r < IPT; r++) {
        const uint32_t p = (uint32_t)(wave * (IPT * 64) + r * 64 + lane);
        if (p < n) dst[p] = val[r];
    }
}

// Segments of up to kTsSmallMax (1 024) pairs: ONE WAVE per tile (round 5, second form: the four-wave workgroups spent their time in barriers --
// six per pass -- and a CU held seven of them; a lone wave needs none, its LDS operations execute in program order, and twenty of
// them share a CU).  Pair p of the segment sits in register p / 64 of lane p % 64; a pass ranks the digits round by round
// (wave_rank), scans the 256 digit counts four per lane, scatters the pairs into LDS and reads them back in order.
template <int IPT, bool GATHER>
__device__ __forceinline__ void tile_sort_wave(const uint2* __restrict__ src, uint32_t* dst, uint32_t n, const uint32_t* __restrict__ dkey,
                                               unsigned long long* s_mask, uint32_t* s_cnt, uint32_t* s_start, uint32_t* s_keys, uint32_t* s_vals)
{
    const int lane = threadIdx.x;
    uint32_t key[IPT], val[IPT], dig[IPT], rnk[IPT];
#pragma unroll
    for (int r = 0; r < IPT; r++) {
        const uint32_t p = (uint32_t)(r * 64 + lane);
        if (GATHER) val[r] = p < n ? dst[p] : 0u;
        else if (p < n) { const uint2 e = src[p]; key[r] = e.x; val[r] = e.y; }
        if (!GATHER && p >= n) { key[r] = 0xffffffffu; val[r] = 0u; }   // padding: behind every real pair, and it stays there
    }
    if (GATHER) {
#pragma unroll
        for (int r = 0; r < IPT; r++) key[r] = (uint32_t)(r * 64 + lane) < n ? dkey[val[r]] : 0xffffffffu;
    }
    uint32_t k_or = 0u, k_and = 0xffffffffu;
#pragma unroll
    for (int r = 0; r < IPT; r++)
        if ((uint32_t)(r * 64 + lane) < n) { k_or |= key[r]; k_and &= key[r]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { k_or |= (uint32_t)__shfl_xor((int)k_or, off, 64); k_and &= (uint32_t)__shfl_xor((int)k_and, off, 64); }
    const uint32_t differ = k_or & ~k_and;   // bits in which two of the tile's keys differ (wave-uniform)
#pragma unroll 1
    for (int shift = 0; shift < 32; shift += 8) {
        if (((differ >> shift) & 0xffu) == 0u) continue;
#pragma unroll
        for (int r = 0; r < IPT; r++) dig[r] = (key[r] >> shift) & 0xffu;
        wave_rank<IPT, 256>(s_mask, s_cnt, dig, rnk, lane);   // s_cnt[d] = the segment's count of digit d (padding in 255, behind the real pairs)
        lds_order();
        {   // exclusive scan of the 256 counts: four digits per lane
            const uint4 c = reinterpret_cast<const uint4*>(s_cnt)[lane];
            const uint32_t sum = c.x + c.y + c.z + c.w;
            const uint32_t ex = wave_inclusive_sum(sum) - sum;
            reinterpret_cast<uint4*>(s_start)[lane] = make_uint4(ex, ex + c.x, ex + c.x + c.y, ex + c.x + c.y + c.z);
        }
        lds_order();
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const uint32_t lp = s_start[dig[r]] + rnk[r];
            s_keys[lp] = key[r]; s_vals[lp] = val[r];
        }
        lds_order();
#pragma unroll
        for (int r = 0; r < IPT; r++) { key[r] = s_keys[r * 64 + lane]; val[r] = s_vals[r * 64 + lane]; }
        lds_order();
    }
#pragma unroll
    for (int r = 0; r < IPT; r++) {
        const uint32_t p = (uint32_t)(r * 64 + lane);
        if (p < n) dst[p] = val[r];
    }
}

// (measured and not kept: segments of 513 ... 1 024 pairs in a launch of their own, so that the sixteen pairs per lane they need do not
//  set the register budget of the short ones -- 8 kB and 60 VGPRs per tile instead of 12 kB and 101: eight batched stage-A models 47 -> 41
//  us, but a model of 300 k Gaussians, most of whose tiles lie in between, 28 -> 39 us for the extra launch)
template <bool GATHER = false>
__global__ __launch_bounds__(64) void k_tile_sort_wave(const uint2* __restrict__ ranges, const uint2* __restrict__ pairs, uint32_t* list, int T,
                                                       const uint32_t* __restrict__ dkey = nullptr)
{
    __shared__ unsigned long long s_mask[256];
    __shared__ __attribute__((aligned(16))) uint32_t s_cnt[256];
    __shared__ __attribute__((aligned(16))) uint32_t s_start[256];
    __shared__ uint32_t s_keys[kTsSmallMax], s_vals[kTsSmallMax];
    const int t = (int)blockIdx.x;
    if (t >= T) return;
    const uint2 rg = ranges[t];
    const uint32_t n = rg.y - rg.x;
    if (n == 0u || n > (uint32_t)kTsSmallMax) return;   // (longer segments: k_tile_sort<1024>)
    const uint2* src = pairs + rg.x;
    uint32_t* dst = list + rg.x;
    if (n <= 64u) tile_sort_wave<1, GATHER>(src, dst, n, dkey, s_mask, s_cnt, s_start, s_keys, s_vals);
    else if (n <= 128u) tile_sort_wave<2, GATHER>(src, dst, n, dkey, s_mask, s_cnt, s_start, s_keys, s_vals);
    else if (n <= 256u) tile_sort_wave<4, GATHER>(src, dst, n, dkey, s_mask, s_cnt, s_start, s_keys, s_vals);
    else if (n <= 512u) tile_sort_wave<8, GATHER>(src, dst, n, dkey, s_mask, s_cnt, s_start, s_keys, s_vals);
    else tile_sort_wave<16, GATHER>(src, dst, n, dkey, s_mask, s_cnt, s_start, s_keys, s_vals);
}

template <int THREADS, bool GATHER = false>
__global__ __launch_bounds__(THREADS) void k_tile_sort(const uint2* __restrict__ ranges, uint2* pairs, uint2* pairs_alt, uint32_t* list, int T,
                                                       const uint32_t* __restrict__ dkey = nullptr)
{
    constexpr int WAVES = THREADS / 64, CAP = THREADS == 256 ? kTsSmallMax : kTsCap;
    static_assert(CAP * 8 == WAVES * 256 * 8, "the pair buffer aliases the mask tables exactly");
    __shared__ unsigned long long s_mask[WAVES][256];   // ranking: XOR masks; between a pass's ranking and the next: the pairs in digit order
    __shared__ uint32_t s_cnt[WAVES][256];
    __shared__ uint32_t s_start[256], s_wsum[WAVES], s_bits[2 * WAVES], s_base[256], s_hist[256];
    uint32_t* const s_keys = reinterpret_cast<uint32_t*>(&s_mask[0][0]);
    uint32_t* const s_vals = s_keys + CAP;
    static_assert(THREADS == 1024, "the short segments are k_tile_sort_wave's");
    // persistent: this workgroup's share of the tiles is t = blockIdx + k gridDim.  Its threads look at one tile each and collect the
    // long ones (one memory round trip for the whole share: walking it tile by tile was 34 dependent loads per workgroup on a batched frame)
    __shared__ uint32_t s_todo[256];
    __shared__ uint32_t s_ntodo;
    if (threadIdx.x == 0) s_ntodo = 0u;
    __syncthreads();
    for (int k = (int)threadIdx.x; (int)blockIdx.x + k * (int)gridDim.x < T; k += THREADS) {
        const int tt = (int)blockIdx.x + k * (int)gridDim.x;
        const uint2 q = ranges[tt];
        if (q.y - q.x > (uint32_t)kTsSmallMax) { const uint32_t slot = atomicAdd(&s_ntodo, 1u); if (slot < 256u) s_todo[slot] = (uint32_t)tt; }
    }
    __syncthreads();
    const uint32_t ntodo = s_ntodo;
    const bool walk_all = ntodo > 256u;   // (more long tiles in this share than the table holds: walk the share)
    const int nloop = walk_all ? (T - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : (int)ntodo;
    for (int it = 0; it < nloop; it++) {
    const int t = walk_all ? (int)blockIdx.x + it * (int)gridDim.x : (int)s_todo[it];
    const uint2 rg = ranges[t];
    const uint32_t n = rg.y - rg.x;
    if (n <= (uint32_t)kTsSmallMax) continue;   // (the other launch's; uniform)
    __syncthreads();   // (the previous tile's LDS is done with)
    if (n <= 2048u) { tile_sort_regs<THREADS, 2, GATHER>(pairs + rg.x, list + rg.x, n, dkey, s_mask, s_cnt, s_keys, s_vals, s_start, s_wsum, s_bits); continue; }
    if (n <= (uint32_t)kTsCap) { tile_sort_regs<THREADS, 4, GATHER>(pairs + rg.x, list + rg.x, n, dkey, s_mask, s_cnt, s_keys, s_vals, s_start, s_wsum, s_bits); continue; }
    // ---- a long segment: the same passes through global memory, a chunk of kTsCap pairs at a time (this workgroup alone reads and
    // writes the segment; its barriers order its own global accesses)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int IPT = kTsCap / THREADS;
    uint2* src = pairs + rg.x;
    uint2* dst = pairs_alt + rg.x;
    if (GATHER) {   // the pairs of this segment, from its indices and the per-Gaussian keys
        for (uint32_t i = (uint32_t)tid; i < n; i += THREADS) { const uint32_t g = list[rg.x + i]; src[i] = make_uint2(dkey[g], g); }
        __syncthreads();
    }
    uint32_t k_or = 0u, k_and = 0xffffffffu;
    for (uint32_t i = (uint32_t)tid; i < n; i += THREADS) { const uint32_t k = src[i].x; k_or |= k; k_and &= k; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { k_or |= (uint32_t)__shfl_xor((int)k_or, off, 64); k_and &= (uint32_t)__shfl_xor((int)k_and, off, 64); }
    if (lane == 0) { s_bits[wave] = k_or; s_bits[WAVES + wave] = k_and; }
    __syncthreads();
    uint32_t differ = 0u;
    {
        uint32_t o = 0u, a = 0xffffffffu;
#pragma unroll
        for (int w = 0; w < WAVES; w++) { o |= s_bits[w]; a &= s_bits[WAVES + w]; }
        differ = o & ~a;
    }
#pragma unroll 1
    for (int shift = 0; shift < 32; shift += 8) {
        if (((differ >> shift) & 0xffu) == 0u) continue;
        __syncthreads();
        if (tid < 256) s_hist[tid] = 0u;                    // digit counts of the whole segment
        __syncthreads();
        for (uint32_t i = (uint32_t)tid; i < n; i += THREADS) atomicAdd(&s_hist[(src[i].x >> shift) & 0xffu], 1u);
        __syncthreads();
        const uint32_t tot = tid < 256 ? s_hist[tid] : 0u;
        const uint32_t ex = block_scan_excl<WAVES>(tot, s_wsum, tid);
        if (tid < 256) s_base[tid] = ex;
        __syncthreads();
        for (uint32_t c0 = 0u; c0 < n; c0 += (uint32_t)kTsCap) {
            const uint32_t m = min((uint32_t)kTsCap, n - c0);
            uint32_t key[IPT], val[IPT], dig[IPT], rnk[IPT];
#pragma unroll
            for (int r = 0; r < IPT; r++) {
                const uint32_t p = (uint32_t)(wave * (IPT * 64) + r * 64 + lane);
                if (p < m) { const uint2 e = src[c0 + p]; key[r] = e.x; val[r] = e.y; dig[r] = (e.x >> shift) & 0xffu; }
                else { key[r] = 0xffffffffu; val[r] = 0u; dig[r] = 255u; }   // (padding: behind the chunk's real pairs of digit 255, never written)
            }
            wave_rank<IPT, 256>(s_mask[wave], s_cnt[wave], dig, rnk, lane);
            __syncthreads();
            uint32_t mine = 0u;
            if (tid < 256) {
#pragma unroll
                for (int w = 0; w < WAVES; w++) { const uint32_t c = s_cnt[w][tid]; s_cnt[w][tid] = mine; mine += c; }
                s_start[tid] = s_base[tid];
            }
            __syncthreads();
            if (tid < 256) s_base[tid] += mine - (tid == 255 ? (uint32_t)kTsCap - m : 0u);   // (the padding was counted in digit 255)
#pragma unroll
            for (int r = 0; r < IPT; r++) {
                const uint32_t p = (uint32_t)(wave * (IPT * 64) + r * 64 + lane);
                if (p < m) dst[s_start[dig[r]] + s_cnt[wave][dig[r]] + rnk[r]] = make_uint2(key[r], val[r]);
            }
            __syncthreads();   // (s_start / s_cnt are rewritten by the next chunk; the stores above are ordered in front of the next pass's loads)
        }
        uint2* tmp = src; src = dst; dst = tmp;
    }
    __syncthreads();
    for (uint32_t i = (uint32_t)tid; i < n; i += THREADS) list[rg.x + i] = src[i].y;
    }
}

// ------------------------------------------------------------------------------------------------
// K7: forward blend.  One 64-lane wave = one workgroup = one 8x8 pixel block; the four waves of a tile are independent: each
// stages the tile's list itself in batches of 64 through LDS (the gathers of the other three hit L2), needs no workgroup
// barrier, and stops as soon as ITS 64 pixels are saturated.  block b: XCD b & 7, slot b >> 3 -> (tile, sub-tile) through
// slot_tile, so a tile's four waves share an XCD (and its L2).  Image state planes and checkpoints: blend_common.h.
// (The tile-per-workgroup kernels, the packed two-pixel kernel and the lane-mask predecessor k_blend_fwd_w of this kernel left the
//  tree in round 5; DESIGN_HISTORY.md has their measurements.)
// "Sign-encoded done": in k_blend_fwd_w `done` is a lane mask the compiler carries in SGPR pairs:
// every iteration opens with xor / and_saveexec / branch on it and closes by merging the lanes that just stopped back in
// -- about sixteen scalar instructions per (wave, instance), as many as the arithmetic (41.5 M SALU next to 46 M VALU
// wave instructions per launch in profiles/r01_pmc_blend.json).  Here a pixel that stops keeps its transmittance with
// the SIGN FLIPPED: T < 0 means done, |T| is the final value.  A done lane then needs no control flow at all:
// T (1 - alpha) is negative, hence below the 1e-4 stop threshold, hence the lane never blends -- the stop test that
// exists anyway masks it.  One divergent region per iteration remains (lanes whose alpha passes), inside it the stop is
// a select, not a branch.  Decisions and results are bit-identical with k_blend_fwd_w for every live lane.
// ------------------------------------------------------------------------------------------------
#ifdef GSR_K6_TIMING   // experiment build only (tools/k6_wave_timing.py): per-wave start / end / placement of the forward blend
__device__ unsigned long long g_k6_dbg[4 * 65536];
__device__ unsigned long long g_k8_dbg[4 * 65536];
__device__ unsigned long long g_k6_cyc[2 * 65536];   // per forward-blend wave: cycles in staging / in the visit loops
__device__ uint32_t g_k6_cnt[2 * 65536];   // per forward-blend wave: visits, taken visits   // the same for the backward blend's workgroups (wave 0)
#endif
#ifndef GSR_FWD_BUFS
#define GSR_FWD_BUFS 2      // staging buffers of a forward-blend wave (round 5, measured: 1 -- a lone wave's LDS operations execute in order, so the next
#endif                      // batch may overwrite the one just visited; 2.5 instead of 5 kB per wave -- the same 96-97 us: LDS does not bound its residency)
constexpr int kFwdBufs = GSR_FWD_BUFS;
template <bool REACH>
__device__ __forceinline__ void blend_fwd_item(const int xcd, const int kslot, float4 (*s_ab)[64], float2 (*s_c)[64],
                                               int W, int H, int tiles_x, int T, const uint2* __restrict__ ranges,
                                               const uint32_t* __restrict__ list, const Splat* __restrict__ splat,
                                               const float* __restrict__ bg, float* __restrict__ out_color,
                                               float* __restrict__ out_depth, float* __restrict__ out_alpha,
                                               float* __restrict__ img, uint32_t* __restrict__ staged4, int interleave,
                                               float* __restrict__ ckpt, int kCkptFirst, int tiles_y, uint16_t* __restrict__ cost_out,
                                               float* __restrict__ out_clamped, const ListCut& cut, const int cut_pass)
{
    // cut_pass (see ListCut): 0 = full lists; 1 = the lists may be cut: a wave that runs out of a cut list with a live pixel flags its
    // tile, every other wave records the depth it needed; 2 = the repair pass: only flagged tiles, on their full lists
    constexpr int NT = 64;
    uint32_t visits = 0u;   // (wave, instance) visits of this item: what the balanced placement of the next render of this view predicts with
#ifdef GSR_K6_TIMING
    const unsigned long long dbg_t0 = wall_clock64();
    uint32_t dbg_visits = 0u, dbg_taken = 0u;   // (wave, instance) visits / visits in which some pixel took the instance
    unsigned long long dbg_stage = 0ull, dbg_loop = 0ull, dbg_mark = 0ull;   // shader-clock cycles: staging (incl. the wait for the gather) / visit loops
#endif
    const int tile = slot_tile(interleave, xcd, kslot >> 2, T, tiles_x);
    const int sub = kslot & 3;
    const int lane = (int)(threadIdx.x & 63u);
    if (tile < 0) { if (cost_out && lane == 0) cost_out[xcd + 8 * kslot] = 0; return; }
    if (cut_pass == 2 && cut.flag[tile] == 0u) return;
    // batched render: T = B tiles_x tiles_y tiles of a tall grid, image `bimg` owns the tile rows [bimg tiles_y, (bimg + 1) tiles_y)
    const int Tl = tiles_x * tiles_y, bimg = tile / Tl, tl = tile - bimg * Tl;
    const int tx = tl % tiles_x, ty = tl / tiles_x;
    const int px = tx * kTile + (sub & 1) * 8 + (lane & 7);
    const int py = ty * kTile + (sub >> 1) * 8 + (lane >> 3);
    // offsets are formed from TILE-relative coordinates (gsr_math.h pixel_rel): the staged mean is relative to the tile's first pixel,
    // remainder included, and the pixel is its column / row inside the tile -- the same numbers the backward blend forms
    const float pxf = (float)((sub & 1) * 8 + (lane & 7)), pyf = (float)((sub >> 1) * 8 + (lane >> 3));
    const float tox = (float)(tx * kTile) - 0.5f * (float)W, toy = (float)(ty * kTile) - 0.5f * (float)H;
    const uint2 rg = ranges[tile];
    const int n = (int)(rg.y - rg.x);
    const int nb = (n + NT - 1) / NT;
    const bool inside = px < W && py < H;
    float Tr = inside ? 1.f : -1.f;            // running transmittance; negative = this pixel is finished
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, Dd = 0.f, Aa = 0.f;
    uint32_t last = 0;
    float4 ra = {0, 0, 0, 0}, rb = ra, rc = ra;
    constexpr float kL2E = 1.4426950408889634f;
    // REACH: the staging lane also runs the exact box test of the tile culling (gsr_math.h box_accept: the minimum of the conic
    // form over the pixel box against 2 ln(255 o) + slack -- never rejects a box in which some pixel is reached) on THIS
    // wave's 8x8 block; the blend loop then visits only the instances whose bit is set.  An instance the tile accepted but
    // this sub-tile cannot see costs three scalar instructions instead of twelve vector ones.
    const float sbx0 = (float)(tx * kTile + (sub & 1) * 8) - 0.5f * (float)W, sby0 = (float)(ty * kTile + (sub >> 1) * 8) - 0.5f * (float)H;
    const float sbx1 = fminf(sbx0 + 7.f, (float)(W - 1) - 0.5f * (float)W), sby1 = fminf(sby0 + 7.f, (float)(H - 1) - 0.5f * (float)H);
    // Staging is a two-level dependent gather (list -> record).  Both levels run AHEAD of their use: the records of batch b + 1 are
    // requested right after batch b is staged and are first touched (box test, pre-scaling, LDS write) at the top of the next
    // iteration, behind this batch's visits; the ids of batch b + 2 are requested at the same time, so the record gather never
    // waits for its addresses.  (Until the end of round 3 the box test sat inside the fetch, so the "prefetch" was consumed where
    // it was issued -- s_waitcnt vmcnt(0) in front of every batch's visits: 115 -> 109 us with the use moved; staging is 7 % of a
    // wave's time afterwards, tools/k6_wave_timing.py.  The same change in the backward blend, whose batches last ~80 us, and its
    // first batch's records requested in front of the pixel-state loads: no change, 2 more VGPRs -- not kept.)
    uint32_t id_nn = 0u;                                   // this lane's id in the batch after the one whose records are in flight
    auto fetch = [&](uint32_t id) {                        // loads only: nothing here may consume them
        const float4* sp = reinterpret_cast<const float4*>(splat + id);
        ra = sp[0]; rb = sp[1]; rc = sp[2];
    };
    // (every lane loads, from a clamped position: a load under a lane condition ends in copies of its result at the join, and
    //  the copies wait for the load right there)
    if (n > 0) {
        fetch(list[rg.x + min(lane, n - 1)]);
        id_nn = list[rg.x + min(NT + lane, n - 1)];
    }
    int batches = 0;
    for (int b = 0; b < nb; b++) {
        const int buf = b & (kFwdBufs - 1);
        if (__all(Tr < 0.f)) break;
        // (round 5, measured and not kept: the six stores moved behind the staging, so that they do not sit between the batch's record
        //  loads and their first use in the wave's in-order vmcnt queue -- 98-100 us against 97)
        if (ckpt && !(b & 1) && (b >> 1) >= kCkptFirst) {   // 128-instance boundary deep in a long list: checkpoint
            float* c = ckpt + ((size_t)(rg.x >> 7) + tile + (b >> 1) - kCkptFirst) * kCkptFloats + sub * 64 + lane;
            c[0] = fabsf(Tr); c[256] = C0; c[512] = C1; c[768] = C2; c[1024] = Dd; c[1280] = Aa;
        }
#ifdef GSR_K6_TIMING
        dbg_mark = __builtin_readcyclecounter();
#endif
        const int cnt = min(NT, n - b * NT);
        // first use of this batch's records (lanes beyond cnt hold stale ones: masked in the ballot, never visited)
        const bool reach_me = REACH && box_accept(make_tile_test(ra.x, ra.y, ra.z, ra.w, rb.x, rb.y), sbx0, sby0, sbx1, sby1);
        {
            const uint32_t lo = __float_as_uint(rc.w);
            ra.x = pixel_rel(ra.x, pixel_lo_x(lo), tox); ra.y = pixel_rel(ra.y, pixel_lo_y(lo), toy);
        }
        ra.z *= -0.5f * kL2E; ra.w *= -kL2E; rb.x *= -0.5f * kL2E;
        s_ab[buf][lane] = ra; s_ab[kFwdBufs + buf][lane] = rb; s_c[buf][lane] = make_float2(rc.x, rc.y);
        const unsigned long long reach = REACH ? __ballot(reach_me && lane < cnt) : 0ull;
        visits += REACH ? (uint32_t)__popcll(reach) : (uint32_t)cnt;
        // the staging area belongs to this wave alone: LDS instructions of one wave execute in issue order, so the broadcast
        // reads below see the writes above -- only the compiler must be kept from reordering them (no s_barrier: the queue
        // kernel runs sixteen independent waves per workgroup)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        batches = b + 1;
#ifdef GSR_K6_TIMING
        { const unsigned long long t = __builtin_readcyclecounter(); dbg_stage += t - dbg_mark; dbg_mark = t; }
#endif
        const int nxt = (b + 1) * NT + lane;
        fetch(id_nn);
        id_nn = list[rg.x + min(nxt + NT, n - 1)];
        auto alpha_of = [&](int j, float& p2) {
            const float4 A = s_ab[buf][j];
            const float2 Bq = *reinterpret_cast<const float2*>(&s_ab[kFwdBufs + buf][j]);   // C', opacity
            const float dx = A.x - pxf, dy = A.y - pyf;
            p2 = fmaf(Bq.x * dy, dy, fmaf(A.w, dy, A.z * dx) * dx);   // log2 of the Gaussian weight
#if defined(__HIP_DEVICE_COMPILE__)
            return fminf(kAlphaMax, Bq.y * __builtin_amdgcn_exp2f(p2));
#else
            return fminf(kAlphaMax, Bq.y * exp2f(p2));
#endif
        };
        auto blend = [&](int j, float p2, float alpha) {
            // No divergent region: the visit is skipped when NO lane's alpha passes (scalar test on the compare masks); otherwise
            // every lane runs the blend with alpha = 0 where its own test failed -- a live pixel then passes the stop test with
            // test_T = T and adds (+-)0 everywhere.  (The exec-masked form cost s_and_saveexec + s_or exec on every visit, taken or
            // not: this loop is bound by instructions issued per wave, scalar ones included.)
            const bool hit = !(p2 > 0.f) && !(alpha < kAlphaMin);
#ifdef GSR_K6_TIMING
            dbg_visits++;
#endif
            if ((__builtin_amdgcn_ballot_w64(!(p2 > 0.f)) & __builtin_amdgcn_ballot_w64(!(alpha < kAlphaMin))) == 0ull) return;
#ifdef GSR_K6_TIMING
            dbg_taken++;
#endif
            const float4 B = s_ab[kFwdBufs + buf][j];
            const float2 C = s_c[buf][j];
            const float am = hit ? alpha : 0.f;
            const float test_T = Tr * (1.f - am);             // negative for a finished pixel: fails the stop test below
            const bool pass = !(test_T < kTStop);
            const float asel = pass ? am : 0.f;               // a lane that does not blend adds (+-)0 to everything below
            const float w = asel * Tr;
            C0 = fmaf(B.w, w, C0); C1 = fmaf(C.x, w, C1); C2 = fmaf(C.y, w, C2);
            Dd = fmaf(B.z, w, Dd); Aa = fmaf(Tr, asel, Aa);   // (k_blend_fwd_w's `A += alpha * T` is contracted to this fma)
            Tr = pass ? test_T : -fabsf(Tr);                  // first failure flips the sign: done, |T| kept
            last = (pass && hit) ? (uint32_t)(b * NT + j + 1) : last;
            // (round 4, measured again at eight waves per SIMD and 0.86 counted vector-pipe activity: the lanes that blend under EXEC
            //  -- `if (hit) { if (pass) {...} else T = -|T| }`, 26 instead of 31 vector instructions per taken visit, three more
            //  branches -- 105 us against 99-101)
        };
        // (round 3: issuing visit k + 1's broadcast reads before visit k's arithmetic -- two register sets, loop unrolled by two, no
        //  copies -- 117 -> 131 us with all three reads prefetched, 135 us with the pre-test fields only: the LDS round trip is not what a
        //  wave waits for here, and the unrolled control flow costs scalar issue slots; the plain loop below stays)
        // (evaluating two instances' alpha before either blend -- two v_exp_f32 in flight -- measured the same: 118 us;
        //  one 16-byte-stride array for the three staged planes, so that a visit needs one address register instead of two:
        //  one v_mov less per visit but 6 instead of 5 kB of LDS per wave -- 26 instead of 32 waves per CU -- 109 -> 113 us)
        if (REACH) {
            for (unsigned long long rm = reach; rm != 0ull;) {
                const int j = (int)__builtin_ctzll(rm);
#if defined(__HIP_DEVICE_COMPILE__)
                asm("s_bitset0_b64 %0, %1" : "+s"(rm) : "s"(j));   // rm &= rm - 1 is s_add_u32 + s_addc_u32 + s_and_b64
#else
                rm &= rm - 1ull;
#endif
                float p2;
                const float a1 = alpha_of(j, p2);
                blend(j, p2, a1);
            }
        } else {
            for (int j = 0; j < cnt; j++) {
                float p2;
                const float a1 = alpha_of(j, p2);
                blend(j, p2, a1);
            }
        }
#ifdef GSR_K6_TIMING
        dbg_loop += __builtin_readcyclecounter() - dbg_mark;
#endif
    }
    if (lane == 0) staged4[tile * 4 + sub] = (uint32_t)min(n, batches * NT);
    if (cost_out && lane == 0) cost_out[xcd + 8 * kslot] = (uint16_t)min(visits, 65535u);
    if (cut_pass) {
        // what this tile needs of its list next time: the depth of the last instance any of its waves staged -- or everything, when a
        // wave reaches the end of the FULL list with a live pixel.  A wave that reaches the end of a CUT list with a live pixel has
        // not seen what it needs: it flags the tile, and the repair pass blends the tile again on its full list.
        const bool live = !__all(Tr < 0.f);
        if (lane == 0) {
            if (cut_pass == 1 && live && cut.tend[tile] < cut.tbase[tile + 1]) {
                if (atomicExch(&cut.flag[tile], 1u) == 0u) atomicAdd(&cut.ctl[2], 1u);
            } else {
                uint32_t k = live ? kCutOpenKey : 0u;
                if (!live && batches > 0) {
                    const int lb = batches - 1, lc = min(NT, n - lb * NT);
                    k = __float_as_uint(s_ab[kFwdBufs + (lb & (kFwdBufs - 1))][lc - 1].z);     // view depth of the last staged instance (positive)
                }
                const uint32_t e = cut.cur[0] < (uint32_t)kVcEntries ? cut.cur[0] : 0u;
                if (k) atomicMax(cut.key + (size_t)e * (size_t)T + tile, k);
            }
        }
    }
#ifdef GSR_K6_TIMING
    if (lane == 0 && xcd + 8 * kslot < 65536) {
        unsigned long long* d = g_k6_dbg + 4 * (size_t)(xcd + 8 * kslot);
        d[0] = dbg_t0; d[1] = wall_clock64();
        d[2] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        d[3] = ((unsigned long long)(uint32_t)n << 32) | (uint32_t)(batches * NT);
        g_k6_cnt[2 * (size_t)(xcd + 8 * kslot)] = dbg_visits; g_k6_cnt[2 * (size_t)(xcd + 8 * kslot) + 1] = dbg_taken;
        g_k6_cyc[2 * (size_t)(xcd + 8 * kslot)] = dbg_stage; g_k6_cyc[2 * (size_t)(xcd + 8 * kslot) + 1] = dbg_loop;
    }
#endif
    if (inside) {
        const size_t Pl = (size_t)W * H, P = Pl * (size_t)(T / Tl), pl = (size_t)py * W + px, pid = (size_t)bimg * Pl + pl;
        const float Tf = fabsf(Tr);
        img[pid] = Tf;
        reinterpret_cast<uint32_t*>(img)[P + pid] = last;
        img[2 * P + pid] = C0; img[3 * P + pid] = C1; img[4 * P + pid] = C2;
        img[5 * P + pid] = Dd; img[6 * P + pid] = Aa;
        float* oc = out_color + (size_t)bimg * 3 * Pl + pl;   // outputs: [B, 3, H, W], [B, 1, H, W]
        const float o0 = C0 + Tf * bg[0], o1 = C1 + Tf * bg[1], o2 = C2 + Tf * bg[2];
        oc[0] = o0;
        oc[Pl] = o1;
        oc[2 * Pl] = o2;
        if (out_clamped) {   // `rendered_image.clamp(0, 1)` (gaussian_model_ht.py:883) written here instead of by a torch launch; NaN stays NaN
            float* cc = out_clamped + (size_t)bimg * 3 * Pl + pl;
            cc[0] = o0 < 0.f ? 0.f : (o0 > 1.f ? 1.f : o0);
            cc[Pl] = o1 < 0.f ? 0.f : (o1 > 1.f ? 1.f : o1);
            cc[2 * Pl] = o2 < 0.f ? 0.f : (o2 > 1.f ? 1.f : o2);
        }
        out_depth[pid] = Dd;
        out_alpha[pid] = Aa;
    }
}


template <bool REACH>
__global__ __launch_bounds__(64) void k_blend_fwd_w6(int W, int H, int tiles_x, int T, const uint2* __restrict__ ranges,
                                                     const uint32_t* __restrict__ list, const Splat* __restrict__ splat,
                                                     const float* __restrict__ bg, float* __restrict__ out_color,
                                                     float* __restrict__ out_depth, float* __restrict__ out_alpha,
                                                     float* __restrict__ img, uint32_t* __restrict__ staged4, int interleave,
                                                     float* __restrict__ ckpt, int kCkptFirst, int tiles_y, const BlendBalance bb,
                                                     float* __restrict__ out_clamped, const ListCut cut = ListCut{}, const int cut_pass = 0)
{
    if (cut_pass == 2 && cut.ctl[2] == 0u) return;      // the repair pass: no tile was flagged -- every wave of the launch leaves here
    // s_a and s_b in ONE array (planes 0/1 = A rows of the two buffers, 2/3 = B rows): a visit's two reads share one address
    // register and differ in the immediate offset
    __shared__ float4 s_ab[2 * kFwdBufs][64];
    __shared__ float2 s_c[kFwdBufs][64];
    // balanced placement (see balance_build): the slot -> item table of this render, and where this item's visits are recorded
    int kslot = (int)(blockIdx.x >> 3);
    uint16_t* cost_out = nullptr;
    if (bb.hdr) {
        const uint32_t* __restrict__ cur = bb.cur;
        const uint16_t* __restrict__ perm = bb.perm;
        const uint32_t e = cur[0];
        if (cut_pass != 2 && cur[1 + (blockIdx.x & 7u)]) {       // this XCD's slice of the table was built (each builder workgroup says so itself); the repair pass: dispatch order
            const int k = (int)perm[blockIdx.x];
            if (k < bb.nslots4) kslot = k;
        }
        if (e < (uint32_t)kVcEntries) cost_out = bb.cost + (size_t)e * bb.items;
    }
    blend_fwd_item<REACH>((int)(blockIdx.x & 7), kslot, s_ab, s_c, W, H, tiles_x, T, ranges, list, splat, bg, out_color, out_depth,
                          out_alpha, img, staged4, interleave, ckpt, kCkptFirst, tiles_y, cost_out, out_clamped, cut, cut_pass);
}

// (round 3, measured with tools/k6_wave_timing.py on the 1 M / 980x545 frame: the 8.6 k waves are all resident at once, eight to
//  a SIMD; they start within 2 us, last 57 us on average (p90 76, max 104), half are gone after 58 us and the SIMDs finish between
//  75 and 106 us -- a SIMD is done when the sum of ITS eight random lists is done, and what is left at the end cannot fill the
//  vector pipe: one wave alone issues an instruction every 5.5 cycles, of any kind, against 2.4 cycles per VALU instruction for a
//  full SIMD.  Pulling the (tile, sub-tile) items from a queue instead -- one 16-wave workgroup per CU, a counter in LDS, ~33 items
//  per workgroup, bit-identical image -- evens the waves out but leaves four per SIMD: 132 us (three: 148, two: 190): per-wave
//  issue, ~55 instructions of all kinds per visit, is what bounds the loop, so residency beats balance.  One counter per XCD in
//  global memory: 210 us -- agent-scope atomics on one address complete about every 0.2 us.  With the counter picked by the hardware's
//  XCC_ID and a WORKGROUP-scope atomic (it then executes in that XCD's L2: `global_atomic_add ... sc0`, no sc1) the pulls cost
//  nothing, single-wave workgroups can be used at any number per SIMD, the image is bit-identical -- and the kernel takes 140 / 131 /
//  127 us at 4 / 5 / 6 persistent waves per SIMD: padding THIS kernel's LDS down to 27 / 24 / 19 / 16 waves per CU gives 123 / 127 /
//  131 / 136 us, and tools/k6_lone_wave.py (uniform lists, no imbalance at all) 48 ns per visit and SIMD at four waves against 43 at
//  eight and 35 in steady state.  Residency beats balance at every point of the curve.  The queue kernel is not kept.
//  Records kept in registers (lane j = instance j) and broadcast with v_readlane_b32 instead of LDS reads -- no LDS, no waits in
//  the visit -- 115 -> 173 us: ten v_readlane per taken visit cost far more vector issue than the two address moves they replace.)

// ------------------------------------------------------------------------------------------------
// Backward prologue (round 4): ONE launch in place of the 48 N-byte memset in front of the backward blend.  Workgroup 0 turns the
// forward's per-sub-tile staged depths into the backward blend's WORK ITEMS -- (tile, [b0, b1) run of 128-instance batches), each
// resuming from the checkpoint the forward left at b0 -- as eight lists, one per XCD of the forward's tile map (a tile's items go
// where its records and pixel planes were last touched), each list ordered by batch index: the first batches of the lists, where
// every pixel is still alive, are the long items and go first.  Every other workgroup zero-fills the per-Gaussian accumulators.
// Until round 3 the blend launched split x Tpad workgroups (33 600 on the 980x545 frame) of which 6 377 found work.
// ------------------------------------------------------------------------------------------------
constexpr size_t kItemHdrBytes = 2048;
struct BwdItemHdr {
    uint32_t count[8];      // items of list x
    uint32_t offset[8];     // first item of list x in the item array
    uint32_t pad[16];
    uint32_t head[8][32];   // queue heads, one 128-byte line each: next item index of list x (device-scope fetch-adds)
};
static_assert(sizeof(BwdItemHdr) <= kItemHdrBytes, "item header");
constexpr int kItemBuckets = 32;   // batch indices 0..30 keep their own bucket of a list, deeper ones share the last

__global__ __launch_bounds__(256) void k_bwd_prologue(float4* __restrict__ gg4, size_t n4, const uint32_t* __restrict__ staged4, int T,
                                                      int tiles_x, int interleave, int tpad, int kCkptFirst, int split_ok,
                                                      BwdItemHdr* __restrict__ hdr, uint2* __restrict__ items, uint32_t list_cap,
                                                      uint32_t first_pull, uint32_t* __restrict__ zero_words, int zero_count)
{
    const int tid = threadIdx.x;
    const int builders = hdr ? 8 : 0;
    if ((int)blockIdx.x >= builders) {
        const size_t nblk = gridDim.x - builders, blk = blockIdx.x - builders;
        const float4 z = {0.f, 0.f, 0.f, 0.f};
        for (size_t i = blk * 256 + tid; i < n4; i += nblk * 256) nt_store4(gg4 + i, z);   // (the blend's atomics are the next to touch these lines)
        return;
    }
    // workgroups 0..7 (the first to start; they are done before the fill is): workgroup x builds list x, in its own region of
    // list_cap items (no cross-workgroup offsets to wait for)
    const int x = (int)blockIdx.x;
    // (prepare in backward: the digit counters that the per-Gaussian kernel behind the blend adds into)
    if (zero_words && x == 0)
        for (int q = tid; q < zero_count; q += 256) zero_words[q] = 0u;
    __shared__ uint32_t s_cnt[kItemBuckets + 1], s_cur[kItemBuckets];
    if (tid <= kItemBuckets) s_cnt[tid] = 0u;
    if (tid < kItemBuckets) s_cur[tid] = 0u;
    __syncthreads();
    auto items_of = [&](int tile) -> int {      // the run [0, kCkptFirst) first, then one item per 128-instance batch
        if (!split_ok) return 1;                // (no checkpoints -- a forward variant that leaves none, or "bwd_split" 1: the whole tile, whatever
                                                //  the staged counters say: those variants do not all write them)
        const uint4 sd = *reinterpret_cast<const uint4*>(staged4 + 4 * tile);
        const int nbs = ((int)max(max(sd.x, sd.y), max(sd.z, sd.w)) + 127) / 128;
        if (nbs == 0) return 0;
        return nbs <= kCkptFirst ? 1 : 1 + nbs - kCkptFirst;
    };
    const int slots = tpad >> 3;
    for (int k = tid; k < slots; k += 256) {
        const int tile = slot_tile(interleave, x, k, T, tiles_x);
        if (tile < 0) continue;
        const int ni = items_of(tile);
        for (int q = 0; q < ni; q++) atomicAdd(&s_cnt[min(q, kItemBuckets - 1)], 1u);
    }
    __syncthreads();
    if (tid == 0) {                              // bucket bases; the total in the last entry
        uint32_t run = 0;
        for (int q = 0; q < kItemBuckets; q++) { const uint32_t c = s_cnt[q]; s_cnt[q] = run; run += c; }
        s_cnt[kItemBuckets] = run;
        hdr->count[x] = min(run, list_cap); hdr->offset[x] = (uint32_t)x * list_cap; hdr->head[x][0] = first_pull;
    }
    __syncthreads();
    uint2* const mine = items + (size_t)x * list_cap;
    for (int k = tid; k < slots; k += 256) {
        const int tile = slot_tile(interleave, x, k, T, tiles_x);
        if (tile < 0) continue;
        const int ni = items_of(tile);
        for (int q = 0; q < ni; q++) {
            const int bk = min(q, kItemBuckets - 1);
            const uint32_t pos = s_cnt[bk] + atomicAdd(&s_cur[bk], 1u);
            // y = first batch of the run; it ends at kCkptFirst (b0 = 0) or after one batch -- bit 31: the tile's only item, runs to the end
            if (pos < list_cap) mine[pos] = make_uint2((uint32_t)tile, (q == 0 ? 0u : (uint32_t)(kCkptFirst + q - 1)) | (ni == 1 ? 0x80000000u : 0u));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K8: backward blend.  Same staging as the forward; front-to-back replay from the stored totals.  The per-(pixel, Gaussian)
// contributions are summed over the lane's pixels, reduced across the wave (transposed DPP butterfly), combined across the
// tile's two waves in LDS and flushed with ONE set of float atomics per (tile, Gaussian) record.
// ggrad record (12 floats / Gaussian): gx gy gA gB gC gop gr gg gb gz - -   (kGG, blend_common.h)
// Packed math:  Each lane owns the two vertically adjacent pixels (x, y0) and (x, y0+1);
// everything per pixel is a float2 so the arithmetic maps to v_pk_mul/add/fma_f32 (two pixels per VALU issue
// slot).  HAS_DA = false drops the depth / alpha-output terms when those upstream gradients are absent (the
// reference never puts loss on them: lambda_depth = 0, /root/reference/arguments/__init__.py:135).  Skip
// decisions are branch-free per pixel (masked alpha and G); only the whole-wave skip is a branch.
// ------------------------------------------------------------------------------------------------
struct BlendBwdArgs {
    int W, H, tiles_x, tiles_y, T, interleave, kCkptFirst, pad;
    const uint2* ranges;
    const uint32_t* list;
    const Splat* splat;
    const float *bg, *img, *g_color, *g_depth, *g_alpha;
    float* ggrad;
    const float* ckpt;
    float* det_part;
    BwdItemHdr* hdr;
    const uint2* items;
};

#ifdef GSR_K8_PHASES   // experiment build only (tools/k8_phases.sh): where a wave of the backward blend spends its time
__device__ unsigned long long g_k8ph[8192 * 8];   // [wave of the last launch][phase]
#define K8_T(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); k8t[k] += now_ - k8t_last; k8t_last = now_; } while (0)
#else
#define K8_T(k) do { } while (0)
#endif
template <bool HAS_DA>
__global__ __launch_bounds__(128) void k_blend_bwd2(BlendBwdArgs args_)
{
#ifdef GSR_K8_PHASES
    unsigned long long k8t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, k8t_last = __builtin_readcyclecounter();
    k8t[6] = k8t_last;
#endif
    constexpr int NT = 128, NV = HAS_DA ? 10 : 9;   // (two waves per workgroup)
    // single staging buffer: a batch is ~10^4 cycles of compute, so the second barrier per batch is free, and the
    // smaller LDS footprint lets more tiles share a CU (latency hiding: waves were 33% in s_waitcnt / barriers)
    // Staged planes, ONE array for the two float4 planes so that a visit's reads share an address register:
    //   s_ab[0] = (x, y, A', B'),  s_ab[1] = (C', opacity, r, g),  s_c = (b, reach bits[, depth, -]).
    // Without the depth / alpha terms the third plane is a float2 and the workgroup's LDS is 10 240 bytes: SIXTEEN workgroups
    // (32 waves, the hardware's limit) per CU would fit instead of fourteen; s_max lives in s_gid's first two words for the same
    // reason.  The kernel's 72 VGPRs still hold it to seven waves per SIMD: forced to 64 (amdgpu_waves_per_eu(8, 8): seven spills)
    // it measured 199 us against 181-190.
    __shared__ float4 s_ab[2][NT];
    __shared__ typename std::conditional<HAS_DA, float4, float2>::type s_c[NT];
    __shared__ uint32_t s_gid[1][NT];
    // ONE row of partials per staged instance: the tile's two waves add theirs into it with ds_add_f32 (each touches a record's nine
    // words once per batch, from nine lanes: conflict-free; 0 + a + b = 0 + b + a, so the arrival order does not matter).  Round 3:
    // a row per wave (4.6 kB more LDS per workgroup: 10 instead of 14 workgroups per CU) was 219-226 us where this is 210-211
    __shared__ float s_part[NT][NV];
    uint32_t* const s_max = &s_gid[0][0];   // [2], only until the staging below (a barrier sits between)
    // PERSISTENT workgroups (round 4): the grid is what the chip holds at once; each workgroup replays work items -- (tile, run of
    // 128-instance batches), built by k_bwd_prologue -- until the eight lists are empty.  Workgroup w starts with item (w >> 3) of
    // list (w & 7) (block b runs on XCD b % 8: the list whose tiles that XCD's L2 saw last -- affinity for speed only), then pulls
    // the next index of that list with a device-scope fetch-add on its head word (sharded per list: a few thousand pulls per launch,
    // spread over its duration), and when the list is exhausted moves on to the next one: placement-independent, and the tail of
    // an unlucky list is shared by everyone.  A head word is peeked with a plain device-scope load first, so the ~28 000 looks at
    // exhausted lists at the end of the launch are loads, not serialised atomics.
    __shared__ uint32_t s_pull;
    constexpr float kL2E = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
#pragma unroll
    for (int k = 0; k < NV; k++) s_part[threadIdx.x][k] = 0.f;   // every flush re-zeroes what it consumed
    int qx = (int)(blockIdx.x & 7u), tried = 0;
    uint32_t qi = blockIdx.x >> 3;
    for (;;) {
    // Nothing but (qx, qi, tried) is carried from item to item.  The kernel's arguments are re-read from the kernarg segment per item
    // (scalar loads through an opaque copy of the segment pointer) and everything derived from them or from the thread index is
    // formed per item (a few dozen scalar / vector instructions per ~80 us item): kept live across the item loop -- thirteen pointers,
    // the frame's dimensions, the hoisted lane constants -- the kernel, 69 SGPRs / 72 VGPRs as a one-item-per-workgroup launch, ran
    // out of scalar registers and took 93 VGPRs (five instead of seven waves per SIMD).
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) BlendBwdArgs* KArgs;
    KArgs ka = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
#else
    const BlendBwdArgs* ka = &args_;
    int tid = (int)threadIdx.x;
#endif
    const int lane = tid & 63, wave = tid >> 6;
    const int part_slot = reduce2_slot<NV>(lane);          // which of a visit's NV totals this lane ends up with (-1: none)
    const uint32_t part_off = (uint32_t)(part_slot < 0 ? 0 : part_slot);
    const int W = ka->W, H = ka->H, tiles_x = ka->tiles_x, tiles_y = ka->tiles_y, T = ka->T, kCkptFirst = ka->kCkptFirst;
    BwdItemHdr* const hdr = ka->hdr;
    const uint2* const __restrict__ ranges = ka->ranges;
    const uint32_t* const __restrict__ list = ka->list;
    const Splat* const __restrict__ splat = ka->splat;
    const float* const __restrict__ bg = ka->bg;
    const float* const __restrict__ img = ka->img;
    const float* const __restrict__ g_color = ka->g_color;
    const float* const __restrict__ g_depth = ka->g_depth;
    const float* const __restrict__ g_alpha = ka->g_alpha;
    float* const __restrict__ ggrad = ka->ggrad;
    const float* const __restrict__ ckpt = ka->ckpt;
    float* const __restrict__ det_part = ka->det_part;
    // ---- next item -------------------------------------------------------------------------------------------------------
    uint32_t qcount = hdr->count[qx];
    while (qi >= qcount) {           // (workgroup-uniform: qi and qx are)
        if (++tried == 8) {
#ifdef GSR_K8_PHASES
            const uint32_t w_ = blockIdx.x * 2u + (threadIdx.x >> 6);
            if ((threadIdx.x & 63) == 0 && w_ < 8192u) { k8t[7] = __builtin_readcyclecounter(); for (int q_ = 0; q_ < 8; q_++) g_k8ph[w_ * 8 + q_] = k8t[q_]; }
#endif
            return;
        }
        qx = (qx + 1) & 7;
        qcount = hdr->count[qx];
        qi = 0xffffffffu;
        if (qcount == 0u) continue;
        if (tid == 0) {
            uint32_t v = __hip_atomic_load(&hdr->head[qx][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v < qcount) v = __hip_atomic_fetch_add(&hdr->head[qx][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_pull = v;
        }
        __syncthreads();
        qi = s_pull;
        __syncthreads();
    }
    const uint2 item = ka->items[hdr->offset[qx] + qi];
    K8_T(0);   // pull / list bookkeeping
#ifdef GSR_K6_TIMING
    const uint32_t dbg_slot = qi < 8192u ? (uint32_t)qx * 8192u + qi : 65536u;   // (probe table: 8 lists x 8 192 items)
    if (tid == 0 && dbg_slot < 65536) { g_k8_dbg[4 * (size_t)dbg_slot] = wall_clock64(); g_k8_dbg[4 * (size_t)dbg_slot + 1] = 0ull; }
#endif
    const int Tl = tiles_x * tiles_y;
    const size_t Pl = (size_t)W * H, P = Pl * (size_t)(T / Tl);
    const float cyf = 0.5f * (float)H;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const int tile = (int)item.x;
    int b0 = (int)(item.y & 0x7fffffffu);
    int b1 = (item.y & 0x80000000u) ? 0x7fffffff : (b0 == 0 ? kCkptFirst : b0 + 1);
    const int bimg = tile / Tl, tl = tile - bimg * Tl;   // batched render: see k_blend_fwd_w6
    const int tx = tl % tiles_x, ty = tl / tiles_x;
    const int px = tx * kTile + (tid & 15);
    const int py0 = ty * kTile + (tid >> 4) * 2;
    // the pixel's column / rows inside the tile; the staged means are relative to the tile's first pixel (gsr_math.h pixel_rel), exactly
    // as the forward blend forms them
    const float pxf = (float)(tid & 15);
    const f2 pyf = {(float)((tid >> 4) * 2), (float)((tid >> 4) * 2 + 1)};
    const uint2 rg = ranges[tile];
    const float* const imgb = img + (size_t)bimg * Pl;                       // this image's slice of every state plane (planes are P apart)
    const float* const g_colorb = g_color ? g_color + (size_t)bimg * 3 * Pl : nullptr;  // upstream gradients: [B, 3, H, W], [B, 1, H, W]
    const float* const g_depthb = g_depth ? g_depth + (size_t)bimg * Pl : nullptr;
    const float* const g_alphab = g_alpha ? g_alpha + (size_t)bimg * Pl : nullptr;
    do {

    // S = <gC, suffix colour> + gD * suffix depth + gA * suffix alpha + T_final <bg, gC>: the only combination of the
    // suffix sums the gradient needs, so ONE running value per pixel replaces five (and the bg term rides along)
    const f2 zero2 = {0.f, 0.f};
    f2 Tt = {1.f, 1.f}, S = zero2;
    f2 gC0 = zero2, gC1 = zero2, gC2 = zero2, gD = zero2, gA = zero2;
    uint32_t ncon[2] = {0u, 0u};
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int py = py0 + p;
        if (px < W && py < H) {
            const size_t pid = (size_t)py * W + px;
            ncon[p] = reinterpret_cast<const uint32_t*>(imgb)[P + pid];
            if (g_colorb) { gC0[p] = g_colorb[pid]; gC1[p] = g_colorb[Pl + pid]; gC2[p] = g_colorb[2 * Pl + pid]; }
            float s = gC0[p] * imgb[2 * P + pid] + gC1[p] * imgb[3 * P + pid] + gC2[p] * imgb[4 * P + pid];
            if (HAS_DA) {
                if (g_depthb) gD[p] = g_depthb[pid];
                if (g_alphab) gA[p] = g_alphab[pid];
                s += gD[p] * imgb[5 * P + pid] + gA[p] * imgb[6 * P + pid];
            }
            S[p] = s + imgb[pid] * (bg0 * gC0[p] + bg1 * gC1[p] + bg2 * gC2[p]);
        }
    }
    uint32_t nmax = max(ncon[0], ncon[1]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, off, 64));
    if (lane == 0) s_max[wave] = nmax;
    __syncthreads();
    const int n = (int)max(s_max[0], s_max[1]);
    const int nw = (int)s_max[wave];
    __syncthreads();   // s_max is s_gid: nobody stages before everyone has read it
    const int nb = (n + NT - 1) / NT;
    b1 = min(b1, nb);
    if (b0 >= b1) break;    // (uniform) the staged depth over-estimated the deepest contributor
    if (b0 > 0) {   // resume from the forward's checkpoint at batch b0: T there, S = what is still to come
        const float* c = ckpt + ((size_t)(rg.x >> 7) + tile + b0 - kCkptFirst) * kCkptFloats;
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int ly = (tid >> 4) * 2 + p, lx = tid & 15;
            const int pi = ((ly >> 3) * 2 + (lx >> 3)) * 64 + (ly & 7) * 8 + (lx & 7);   // k_blend_fwd_w's (sub-tile, lane)
            if (ncon[p] > (uint32_t)(b0 * NT)) {   // the pixel was still blending at this boundary: its wave wrote the slot
                Tt[p] = c[pi];
                float s = gC0[p] * c[256 + pi] + gC1[p] * c[512 + pi] + gC2[p] * c[768 + pi];
                if (HAS_DA) s += gD[p] * c[1024 + pi] + gA[p] * c[1280 + pi];
                S[p] -= s;
            } else {                                // finished earlier: takes no part here (and its slot may be unwritten)
                Tt[p] = 0.f; S[p] = 0.f; ncon[p] = 0u;
            }
        }
    }

    K8_T(1);   // pixel state in (planes, upstream gradients, checkpoint) + the two barriers
    float4 ra = {0, 0, 0, 0}, rb = ra, rc = ra;
    uint32_t rg_id = 0;
    // the staged copy carries the conic pre-multiplied for the exponent in base 2, exactly as k_blend_fwd_w stages it
    // (A' = -log2(e)/2 A, B' = -log2(e) B, C' = -log2(e)/2 C), and log2(G) is evaluated with the same fma nesting, so
    // forward and backward agree on every skip decision bit for bit
    // Per staged instance the staging thread also decides, once, which of the two waves (16x8 pixel halves of the tile)
    // the Gaussian can reach at all: the exact box test of the tile culling (gsr_math.h box_accept: minimum of the conic
    // form over the half's pixel box against 2 ln(255 o) + slack) on each half.  A wave skips an instance whose bit is
    // clear for the cost of one LDS read instead of evaluating 128 alphas to find that none passes (36 % of the
    // (wave, instance) iterations were such misses).  rc.w (Splat::tiles, unused by the blend) carries the two bits.
    const float hbx0 = (float)(tx * kTile) - 0.5f * (float)W, hbx1 = fminf(hbx0 + (float)(kTile - 1), (float)(W - 1) - 0.5f * (float)W);
    const float hby0 = (float)(ty * kTile) - cyf, hbyL = (float)(H - 1) - cyf;
    // Round 5: the two-level gather (list -> record) of a batch is REQUESTED one batch ahead and first TOUCHED at the top of the batch that
    // stages it.  Until now `stage(next)` sat behind the barrier with its arithmetic (the box tests of the reach bits) right behind its
    // loads, so hipcc put `s_waitcnt vmcnt(0)` -- two dependent memory round trips, index then record -- in front of every batch's
    // visits (the forward blend had the same fault until round 3).  Now the record of batch b + 1 is requested with an index that was
    // itself requested during batch b - 1, and nothing between the requests and the next batch's top reads either.
    uint32_t id_nxt = 0u;     // list entry of this thread's instance in the batch after the one in (ra, rb, rc)
    bool pend = false;        // (ra, rb, rc) hold a raw record that stage_finish has not yet turned into its staged form
    // (EVERY lane requests, a lane without an instance from a clamped position: behind a divergent branch the loaded registers meet the
    //  old ones at a join, hipcc copies them there, and the copy waits for the load)
    auto stage_request = [&](uint32_t id, bool want) {
        rg_id = id;
        // three whole 16-byte loads into three aligned register quads: left to itself hipcc trims the record's unused words away and
        // lands a lone dword in a register whose neighbour is the broadcast operand of a packed instruction of the visit -- which then
        // waits for the load (`v_pk_fma v[38:39], v[28:29], ...` with the load's v29 in flight)
        typedef float vf4 __attribute__((ext_vector_type(4)));
        const vf4* sp = reinterpret_cast<const vf4*>(splat + id);
        const vf4 q0 = sp[0], q1 = sp[1], q2 = sp[2];
        ra = make_float4(q0.x, q0.y, q0.z, q0.w); rb = make_float4(q1.x, q1.y, q1.z, q1.w); rc = make_float4(q2.x, q2.y, q2.z, q2.w);
        pend = want;
    };
    auto list_at = [&](int idx) -> uint32_t { return list[rg.x + (uint32_t)min(idx, n - 1)]; };
    auto stage_finish = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
        // the record's two words the blend does not read stay "used" until here: a register the allocator takes for dead it hands to the
        // visit loop as a temporary, and writing it waits for the load that is still filling its quad
        asm volatile("" :: "v"(rb.z), "v"(rc.z));
#endif
        const TileTest tt = make_tile_test(ra.x, ra.y, ra.z, ra.w, rb.x, rb.y);
        uint32_t fl = 0u;
        if (hby0 <= hbyL) fl |= box_accept(tt, hbx0, hby0, hbx1, fminf(hby0 + 7.f, hbyL)) ? 1u : 0u;
        if (hby0 + 8.f <= hbyL) fl |= box_accept(tt, hbx0, hby0 + 8.f, hbx1, fminf(hby0 + 15.f, hbyL)) ? 2u : 0u;
        {
            const uint32_t lo = __float_as_uint(rc.w);   // (the record's `tiles` word: the remainders of the mean)
            ra.x = pixel_rel(ra.x, pixel_lo_x(lo), hbx0); ra.y = pixel_rel(ra.y, pixel_lo_y(lo), hby0);
        }
        rc.w = __uint_as_float(fl);
        ra.z *= -0.5f * kL2E; ra.w *= -kL2E; rb.x *= -0.5f * kL2E;
        pend = false;
    };
    stage_request(list_at(b0 * NT + tid), b0 * NT + tid < n);
    id_nxt = list_at((b0 + 1) * NT + tid);
    for (int b = b0; b < b1; b++) {
        const int buf = 0;
        if (b > b0) __syncthreads();   // everyone is done reading the previous batch (and its flush read s_gid)
        if (pend) stage_finish();      // first touch of the record requested a batch ago
        s_ab[0][tid] = ra; s_ab[1][tid] = make_float4(rb.x, rb.y, rb.w, rc.x); s_gid[buf][tid] = rg_id;
        if constexpr (HAS_DA) s_c[tid] = make_float4(rc.y, rc.w, rb.z, 0.f); else s_c[tid] = make_float2(rc.y, rc.w);
        __syncthreads();
        stage_request(id_nxt, (b + 1) * NT + tid < n && b + 1 < b1);   // (its index arrived during the last batch)
        id_nxt = list_at((b + 2) * NT + tid);
        K8_T(2);   // staging: wait for the record, box tests, LDS, barriers, next requests
        const int cnt = min(NT, n - b * NT);
        // a wave only walks as far as ITS pixels' last contributor (the tile-wide n bounds the staging and the barriers)
        const int cntw = min(cnt, nw - b * NT);
        // the reach bits of the batch as two wave-uniform 64-bit masks: the loop below visits set bits only, so an
        // instance this half cannot reach costs nothing at all
        const uint32_t wbit = 1u << wave;
        const bool r0 = lane < cntw && (__float_as_uint(s_c[lane].y) & wbit);
        const bool r1 = lane + 64 < cntw && (__float_as_uint(s_c[lane + 64].y) & wbit);
        const unsigned long long reach[2] = {__ballot(r0), __ballot(r1)};
#pragma unroll 1
        for (int half = 0; half < 2; half++)
        for (unsigned long long rm = reach[half]; rm != 0ull; rm &= rm - 1ull) {
            const int j = half * 64 + (int)__builtin_ctzll(rm);
            const float4 A = s_ab[0][j], B = s_ab[1][j];   // (manual LDS prefetch measured slower)
            const auto C = s_c[j];
            const uint32_t idx = (uint32_t)(b * NT + j + 1);
            const float ca = A.z, cb = A.w, cc = B.x, op = B.y, cr = B.z, cg = B.w, cbl = C.x;   // ca/cb/cc: A', B', C'
            float zd = 0.f;
            if constexpr (HAS_DA) zd = C.z;
            const float dx = A.x - pxf;
            const f2 dy = A.y - pyf;
            const float adx = ca * dx;
            const f2 cdy = cc * dy;
            const f2 p2 = fma2(cdy, dy, fma2(f2{cb, cb}, dy, f2{adx, adx}) * dx);   // log2 of the Gaussian weight (= k_blend_fwd_w)
#if defined(__HIP_DEVICE_COMPILE__)
            f2 G = {__builtin_amdgcn_exp2f(p2.x), __builtin_amdgcn_exp2f(p2.y)};
#else
            f2 G = {exp2f(p2.x), exp2f(p2.y)};
#endif
            f2 alpha = op * G;   // clamped to kAlphaMax only once the visit is taken (the 1/255 test below reads the same either way)
            const bool c0p = !(p2.x > 0.f), c0a = !(alpha.x < kAlphaMin), c0n = idx <= ncon[0];
            const bool c1p = !(p2.y > 0.f), c1a = !(alpha.y < kAlphaMin), c1n = idx <= ncon[1];
            const bool v0 = c0p && c0a && c0n, v1 = c1p && c1a && c1n;
            // wave-uniform test on the compare masks themselves: a ballot of the combined bool is lowered to v_cndmask + v_cmp
            // (two vector instructions per visit, taken or not); ballots of the single compares ARE the compares' SGPR results
            const unsigned long long any01 =
                (__builtin_amdgcn_ballot_w64(c0p) & __builtin_amdgcn_ballot_w64(c0a) & __builtin_amdgcn_ballot_w64(c0n)) |
                (__builtin_amdgcn_ballot_w64(c1p) & __builtin_amdgcn_ballot_w64(c1a) & __builtin_amdgcn_ballot_w64(c1n));
            if (any01 != 0ull) {
                // mask G, then alpha from the masked G: two selects, a packed multiply and the clamp instead of four selects here
                // and the clamp on every visit
                G.x = v0 ? G.x : 0.f; G.y = v1 ? G.y : 0.f;
                alpha = op * G;
                alpha.x = fminf(kAlphaMax, alpha.x); alpha.y = fminf(kAlphaMax, alpha.y);
                const f2 w = alpha * Tt;
                f2 gc = gC0 * cr + gC1 * cg + gC2 * cbl;          // <gC, colour of this Gaussian> (+ depth / alpha terms)
                if (HAS_DA) gc += gD * zd + gA;
                S -= gc * w;
                const f2 om = 1.f - alpha;
                // one reciprocal for the two pixels: 1/a = b/(ab), 1/b = a/(ab)  (om >= 0.01: the product cannot underflow; v_rcp_f32
                // runs at a quarter of the VALU rate, two of them were 8 % of this loop's issue time)
                const float rab = fast_rcp(om.x * om.y);
                const f2 inv = f2{om.y, om.x} * rab;
                const f2 dLda = gc * Tt - S * inv;
                const f2 dLdpow = G * (op * dLda);
                // The five geometric gradients are linear in the moments of dLdpow over the pixels:
                //   gx = B' my + 2 A' mx, gy = 2 C' my + B' mx, gA = -mxx / 2, gB = -mxy, gC = -myy / 2
                // with mx = sum dLdpow dx, my = sum dLdpow dy, mxx = sum dLdpow dx^2, ...  Only the moments are formed and
                // reduced here (five packed products); the per-record combination happens once, in the flush.
                // dx is shared by a lane's two pixels (same column), so the x moments come from the folded sums with one
                // plain multiply each (mx = dx sum dLdpow, mxx = dx mx, mxy = dx my); the per-pixel products that remain are
                // folded with mul + fma (two full-rate ops) instead of a packed multiply and an add (4.3 + 2.4 cycles)
                const f2 t_my = dLdpow * dy;
                float v[10];
                v[0] = (dLdpow.x + dLdpow.y) * dx;
                v[1] = t_my.x + t_my.y;
                v[2] = v[0] * dx;
                v[3] = v[1] * dx;
                v[4] = __builtin_fmaf(t_my.y, dy.y, t_my.x * dy.x);
                v[5] = __builtin_fmaf(G.y, dLda.y, G.x * dLda.x);
                v[6] = __builtin_fmaf(w.y, gC0.y, w.x * gC0.x);
                v[7] = __builtin_fmaf(w.y, gC1.y, w.x * gC1.x);
                v[8] = __builtin_fmaf(w.y, gC2.y, w.x * gC2.x);
                if (HAS_DA) v[9] = __builtin_fmaf(w.y, gD.y, w.x * gD.x);
                Tt *= om;
                // (measured: finishing the reduction with ds_add_f32 from the row leaders is 1.7x SLOWER -- four lanes on one
                //  address serialise; the transposed DPP reduction below halves the VALU cost instead)
                const float t = wave_reduce_transposed2<NV>(v, lane);
                // nine (ten) lanes of the first row hold one total each (reduce2_slot).  The row offset is a SCALAR product (j is
                // wave-uniform); left to itself the compiler folds it into a v_mad_u64_u32 per visit.
                uint32_t row;
#if defined(__HIP_DEVICE_COMPILE__)
                asm("s_mul_i32 %0, %1, %2" : "=s"(row) : "s"(j), "n"(NV));
#else
                row = (uint32_t)j * NV;
#endif
                if (part_slot >= 0) atomicAdd(&(&s_part[0][0])[part_off + row], t);
            }
        }
        K8_T(3);   // visits
        __syncthreads();
        K8_T(4);   // barrier behind the visits (waiting for the other wave)
        // flush: 16 lanes per Gaussian, lane r adds component r, so one atomic instruction touches 8 records of
        // 9-10 CONSECUTIVE floats (8 cache lines per wave instruction) instead of 64 scattered records -- device-scope
        // float atomics are fabric transactions on this chip, and they were 27% of this kernel when issued one
        // component at a time per lane
        {
            const int r = tid & 15, q = tid >> 4;   // 8 groups of 16 lanes
            for (int jj = q; jj < cnt; jj += NT / 16) {
                if (r < NV) {
                    float v = s_part[jj][r];
                    if (r < 2) {        // moments -> d/d(pixel-space mean): this lane also needs the OTHER first-order moment
                        const float mo = s_part[jj][r ^ 1];
                        const float cb = s_ab[0][jj].w;                                // B'
                        const float c2 = 2.f * (r == 0 ? s_ab[0][jj].z : s_ab[1][jj].x);   // 2 A' (gx) or 2 C' (gy)
                        v = fmaf(cb, mo, c2 * v);
                    } else if (r < 5) v *= (r == 3 ? -1.f : -0.5f);                       // second moments -> conic gradients
                    __builtin_amdgcn_wave_barrier();   // every lane of the group has read both first moments before any is cleared
                    s_part[jj][r] = 0.f;   // ready for the next batch (its writers sit behind a barrier)
                    // deterministic debug mode: the (tile, instance) partial goes to its own slot, k_det_reduce sums a
                    // Gaussian's slots in list order afterwards; default: one coalesced atomic per record row
                    if (det_part) det_part[((size_t)rg.x + (size_t)b * NT + jj) * kDetStride + r] = r < 2 ? v * kLn2 : v;
                    else if (v != 0.f) atomicAdd(ggrad + (size_t)s_gid[buf][jj] * kGG + r, r < 2 ? v * kLn2 : v);
                }
            }
        }
        K8_T(5);   // flush
    }
#ifdef GSR_K6_TIMING
    if (threadIdx.x == 0 && dbg_slot < 65536) {
        unsigned long long* d = g_k8_dbg + 4 * (size_t)dbg_slot;
        d[1] = wall_clock64();
        d[2] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        d[3] = ((unsigned long long)(uint32_t)n << 32) | (uint32_t)((b1 - b0) * NT);
    }
#endif
    } while (0);
    // ---- pull the next index of this list ----------------------------------------------------------------------------------
    // (at the END of the item: issued at its top -- to hide the round trip behind the item -- the returning atomic sat at the head
    //  of the wave's in-order vmcnt queue, and at t = 0 all 3 584 workgroups pull at once, ~450 per head word: the wave of thread 0
    //  could not consume its own pixel loads until its pull had come back, 220-230 us per launch against 180-190 without)
    __syncthreads();   // (s_pull: everyone has read the previous value)
    if (tid == 0) s_pull = __hip_atomic_fetch_add(&hdr->head[qx][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    qi = s_pull;
    tried = 0;
    }
}

// K8s (round 5): the same replay with ONE pixel per lane and FOUR waves per tile -- wave w owns the 8x8 block (w & 1, w >> 1) of the tile, the
// forward blend's sub-tile w, lane = its lane -- for scenes of small splats: a staged instance carries four reach bits instead of two and a
// wave visits it only when its own 64 pixels can see it (a pixel-sized splat reaches one or two of the four blocks, but always a whole
// 16x8 half).  Scalar arithmetic (nothing to pack), the same reduction, staging, work items, checkpoints and flush as k_blend_bwd2.
template <bool HAS_DA>
__global__ __launch_bounds__(256) void k_blend_bwd1(BlendBwdArgs args_)
{
    constexpr int NT = 128, NTH = 256, NV = HAS_DA ? 10 : 9;   // (instances per staged batch; threads = four waves per workgroup)
    // (staging planes, partial rows, work items and flush: see k_blend_bwd2)
    __shared__ float4 s_ab[2][NT];
    __shared__ typename std::conditional<HAS_DA, float4, float2>::type s_c[NT];
    __shared__ uint32_t s_gid[1][NT];
    __shared__ float s_part[NT][NV];
    uint32_t* const s_max = &s_gid[0][0];   // [4], only until the staging below (a barrier sits between)
    __shared__ uint32_t s_pull;
    constexpr float kL2E = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
#pragma unroll
    for (int k = 0; k < NV; k++) if (threadIdx.x < NT) s_part[threadIdx.x][k] = 0.f;   // every flush re-zeroes what it consumed
    int qx = (int)(blockIdx.x & 7u), tried = 0;
    uint32_t qi = blockIdx.x >> 3;
    for (;;) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) BlendBwdArgs* KArgs;
    KArgs ka = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
#else
    const BlendBwdArgs* ka = &args_;
    int tid = (int)threadIdx.x;
#endif
    const int lane = tid & 63, wave = tid >> 6;
    const int part_slot = reduce2_slot<NV>(lane);          // which of a visit's NV totals this lane ends up with (-1: none)
    const uint32_t part_off = (uint32_t)(part_slot < 0 ? 0 : part_slot);
    const int W = ka->W, H = ka->H, tiles_x = ka->tiles_x, tiles_y = ka->tiles_y, T = ka->T, kCkptFirst = ka->kCkptFirst;
    BwdItemHdr* const hdr = ka->hdr;
    const uint2* const __restrict__ ranges = ka->ranges;
    const uint32_t* const __restrict__ list = ka->list;
    const Splat* const __restrict__ splat = ka->splat;
    const float* const __restrict__ bg = ka->bg;
    const float* const __restrict__ img = ka->img;
    const float* const __restrict__ g_color = ka->g_color;
    const float* const __restrict__ g_depth = ka->g_depth;
    const float* const __restrict__ g_alpha = ka->g_alpha;
    float* const __restrict__ ggrad = ka->ggrad;
    const float* const __restrict__ ckpt = ka->ckpt;
    float* const __restrict__ det_part = ka->det_part;
    // ---- next item -------------------------------------------------------------------------------------------------------
    uint32_t qcount = hdr->count[qx];
    while (qi >= qcount) {           // (workgroup-uniform: qi and qx are)
        if (++tried == 8) {
            return;
        }
        qx = (qx + 1) & 7;
        qcount = hdr->count[qx];
        qi = 0xffffffffu;
        if (qcount == 0u) continue;
        if (tid == 0) {
            uint32_t v = __hip_atomic_load(&hdr->head[qx][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v < qcount) v = __hip_atomic_fetch_add(&hdr->head[qx][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_pull = v;
        }
        __syncthreads();
        qi = s_pull;
        __syncthreads();
    }
    const uint2 item = ka->items[hdr->offset[qx] + qi];
    const int Tl = tiles_x * tiles_y;
    const size_t Pl = (size_t)W * H, P = Pl * (size_t)(T / Tl);
    const float cyf = 0.5f * (float)H;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const int tile = (int)item.x;
    int b0 = (int)(item.y & 0x7fffffffu);
    int b1 = (item.y & 0x80000000u) ? 0x7fffffff : (b0 == 0 ? kCkptFirst : b0 + 1);
    const int bimg = tile / Tl, tl = tile - bimg * Tl;   // batched render: see k_blend_fwd_w6
    const int tx = tl % tiles_x, ty = tl / tiles_x;
    const int sbx = (wave & 1) * 8, sby = (wave >> 1) * 8;   // this wave's 8x8 block inside the tile (= the forward blend's sub-tile `wave`)
    const int px = tx * kTile + sbx + (lane & 7);
    const int py = ty * kTile + sby + (lane >> 3);
    // the pixel's column / row inside the tile; the staged means are relative to the tile's first pixel (gsr_math.h pixel_rel), exactly
    // as the forward blend forms them
    const float pxf = (float)(sbx + (lane & 7)), pyf = (float)(sby + (lane >> 3));
    const uint2 rg = ranges[tile];
    const float* const imgb = img + (size_t)bimg * Pl;                       // this image's slice of every state plane (planes are P apart)
    const float* const g_colorb = g_color ? g_color + (size_t)bimg * 3 * Pl : nullptr;  // upstream gradients: [B, 3, H, W], [B, 1, H, W]
    const float* const g_depthb = g_depth ? g_depth + (size_t)bimg * Pl : nullptr;
    const float* const g_alphab = g_alpha ? g_alpha + (size_t)bimg * Pl : nullptr;
    do {

    // S = <gC, suffix colour> + gD * suffix depth + gA * suffix alpha + T_final <bg, gC>: the only combination of the
    // suffix sums the gradient needs, so ONE running value per pixel replaces five (and the bg term rides along)
    float Tt = 1.f, S = 0.f, gC0 = 0.f, gC1 = 0.f, gC2 = 0.f, gD = 0.f, gA = 0.f;
    uint32_t ncon = 0u;
    if (px < W && py < H) {
        const size_t pid = (size_t)py * W + px;
        ncon = reinterpret_cast<const uint32_t*>(imgb)[P + pid];
        if (g_colorb) { gC0 = g_colorb[pid]; gC1 = g_colorb[Pl + pid]; gC2 = g_colorb[2 * Pl + pid]; }
        float s = gC0 * imgb[2 * P + pid] + gC1 * imgb[3 * P + pid] + gC2 * imgb[4 * P + pid];
        if (HAS_DA) {
            if (g_depthb) gD = g_depthb[pid];
            if (g_alphab) gA = g_alphab[pid];
            s += gD * imgb[5 * P + pid] + gA * imgb[6 * P + pid];
        }
        S = s + imgb[pid] * (bg0 * gC0 + bg1 * gC1 + bg2 * gC2);
    }
    uint32_t nmax = ncon;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, off, 64));
    if (lane == 0) s_max[wave] = nmax;
    __syncthreads();
    const int n = (int)max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
    const int nw = (int)s_max[wave];
    __syncthreads();   // s_max is s_gid: nobody stages before everyone has read it
    const int nb = (n + NT - 1) / NT;
    b1 = min(b1, nb);
    if (b0 >= b1) break;    // (uniform) the staged depth over-estimated the deepest contributor
    if (b0 > 0) {   // resume from the forward's checkpoint at batch b0: T there, S = what is still to come
        const float* c = ckpt + ((size_t)(rg.x >> 7) + tile + b0 - kCkptFirst) * kCkptFloats;
        const int pi = wave * 64 + lane;   // k_blend_fwd_w's (sub-tile, lane): this very wave and lane
        if (ncon > (uint32_t)(b0 * NT)) {   // the pixel was still blending at this boundary: its wave wrote the slot
            Tt = c[pi];
            float s = gC0 * c[256 + pi] + gC1 * c[512 + pi] + gC2 * c[768 + pi];
            if (HAS_DA) s += gD * c[1024 + pi] + gA * c[1280 + pi];
            S -= s;
        } else {                            // finished earlier: takes no part here (and its slot may be unwritten)
            Tt = 0.f; S = 0.f; ncon = 0u;
        }
    }

    float4 ra = {0, 0, 0, 0}, rb = ra, rc = ra;
    uint32_t rg_id = 0;
    // (the staged copy: as k_blend_bwd2; rc.w carries FOUR reach bits, one per 8x8 block: the box test of the forward blend's sub-tiles)
    const float hbx0 = (float)(tx * kTile) - 0.5f * (float)W, hbx1 = fminf(hbx0 + (float)(kTile - 1), (float)(W - 1) - 0.5f * (float)W);
    const float hby0 = (float)(ty * kTile) - cyf, hbyL = (float)(H - 1) - cyf, hbxL = (float)(W - 1) - 0.5f * (float)W;
    (void)hbx1;
    // (records requested one batch ahead, first touched at the top of the batch that stages them: see k_blend_bwd2)
    uint32_t id_nxt = 0u;     // list entry of this thread's instance in the batch after the one in (ra, rb, rc)
    bool pend = false;        // (ra, rb, rc) hold a raw record that stage_finish has not yet turned into its staged form
    // (EVERY lane requests, a lane without an instance from a clamped position: behind a divergent branch the loaded registers meet the
    //  old ones at a join, hipcc copies them there, and the copy waits for the load)
    auto stage_request = [&](uint32_t id, bool want) {
        rg_id = id;
        // three whole 16-byte loads into three aligned register quads: left to itself hipcc trims the record's unused words away and
        // lands a lone dword in a register whose neighbour is the broadcast operand of a packed instruction of the visit -- which then
        // waits for the load (`v_pk_fma v[38:39], v[28:29], ...` with the load's v29 in flight)
        typedef float vf4 __attribute__((ext_vector_type(4)));
        const vf4* sp = reinterpret_cast<const vf4*>(splat + id);
        const vf4 q0 = sp[0], q1 = sp[1], q2 = sp[2];
        ra = make_float4(q0.x, q0.y, q0.z, q0.w); rb = make_float4(q1.x, q1.y, q1.z, q1.w); rc = make_float4(q2.x, q2.y, q2.z, q2.w);
        pend = want;
    };
    auto list_at = [&](int idx) -> uint32_t { return list[rg.x + (uint32_t)min(idx, n - 1)]; };
    auto stage_finish = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
        // the record's two words the blend does not read stay "used" until here: a register the allocator takes for dead it hands to the
        // visit loop as a temporary, and writing it waits for the load that is still filling its quad
        asm volatile("" :: "v"(rb.z), "v"(rc.z));
#endif
        const TileTest tt = make_tile_test(ra.x, ra.y, ra.z, ra.w, rb.x, rb.y);
        uint32_t fl = 0u;
#pragma unroll
        for (int sb = 0; sb < 4; sb++) {   // block sb = (sb & 1, sb >> 1): its pixel box clipped to the frame, as k_blend_fwd_w6 tests its sub-tile
            const float x0 = hbx0 + (float)((sb & 1) * 8), y0 = hby0 + (float)((sb >> 1) * 8);
            if (x0 <= hbxL && y0 <= hbyL) fl |= box_accept(tt, x0, y0, fminf(x0 + 7.f, hbxL), fminf(y0 + 7.f, hbyL)) ? (1u << sb) : 0u;
        }
        {
            const uint32_t lo = __float_as_uint(rc.w);   // (the record's `tiles` word: the remainders of the mean)
            ra.x = pixel_rel(ra.x, pixel_lo_x(lo), hbx0); ra.y = pixel_rel(ra.y, pixel_lo_y(lo), hby0);
        }
        rc.w = __uint_as_float(fl);
        ra.z *= -0.5f * kL2E; ra.w *= -kL2E; rb.x *= -0.5f * kL2E;
        pend = false;
    };
    const bool stager = tid < NT;   // (waves 0 and 1 stage the batch of 128; all four visit it)
    if (stager) {
        stage_request(list_at(b0 * NT + tid), b0 * NT + tid < n);
        id_nxt = list_at((b0 + 1) * NT + tid);
    }
    for (int b = b0; b < b1; b++) {
        const int buf = 0;
        if (b > b0) __syncthreads();   // everyone is done reading the previous batch (and its flush read s_gid)
        if (pend) stage_finish();      // first touch of the record requested a batch ago
        if (stager) {
            s_ab[0][tid] = ra; s_ab[1][tid] = make_float4(rb.x, rb.y, rb.w, rc.x); s_gid[buf][tid] = rg_id;
            if constexpr (HAS_DA) s_c[tid] = make_float4(rc.y, rc.w, rb.z, 0.f); else s_c[tid] = make_float2(rc.y, rc.w);
        }
        __syncthreads();
        if (stager) {
            stage_request(id_nxt, (b + 1) * NT + tid < n && b + 1 < b1);   // (its index arrived during the last batch)
            id_nxt = list_at((b + 2) * NT + tid);
        }
        const int cnt = min(NT, n - b * NT);
        // a wave only walks as far as ITS pixels' last contributor (the tile-wide n bounds the staging and the barriers)
        const int cntw = min(cnt, nw - b * NT);
        // the reach bits of the batch as two wave-uniform 64-bit masks: the loop below visits set bits only, so an
        // instance this half cannot reach costs nothing at all
        const uint32_t wbit = 1u << wave;
        const bool r0 = lane < cntw && (__float_as_uint(s_c[lane].y) & wbit);
        const bool r1 = lane + 64 < cntw && (__float_as_uint(s_c[lane + 64].y) & wbit);
        const unsigned long long reach[2] = {__ballot(r0), __ballot(r1)};
#pragma unroll 1
        for (int half = 0; half < 2; half++)
        for (unsigned long long rm = reach[half]; rm != 0ull; rm &= rm - 1ull) {
            const int j = half * 64 + (int)__builtin_ctzll(rm);
            const float4 A = s_ab[0][j], B = s_ab[1][j];   // (manual LDS prefetch measured slower)
            const auto C = s_c[j];
            const uint32_t idx = (uint32_t)(b * NT + j + 1);
            const float ca = A.z, cb = A.w, cc = B.x, op = B.y, cr = B.z, cg = B.w, cbl = C.x;   // ca/cb/cc: A', B', C'
            float zd = 0.f;
            if constexpr (HAS_DA) zd = C.z;
            const float dx = A.x - pxf, dy = A.y - pyf;
            const float p2 = __builtin_fmaf(cc * dy, dy, __builtin_fmaf(cb, dy, ca * dx) * dx);   // log2 of the Gaussian weight (= k_blend_fwd_w, = k_blend_bwd2 per component)
#if defined(__HIP_DEVICE_COMPILE__)
            float G = __builtin_amdgcn_exp2f(p2);
#else
            float G = exp2f(p2);
#endif
            float alpha = op * G;   // clamped to kAlphaMax only once the visit is taken (the 1/255 test below reads the same either way)
            const bool c0p = !(p2 > 0.f), c0a = !(alpha < kAlphaMin), c0n = idx <= ncon;
            const bool v0 = c0p && c0a && c0n;
            // (wave-uniform test on the compare masks themselves: see k_blend_bwd2)
            const unsigned long long any0 = __builtin_amdgcn_ballot_w64(c0p) & __builtin_amdgcn_ballot_w64(c0a) & __builtin_amdgcn_ballot_w64(c0n);
            if (any0 != 0ull) {
                G = v0 ? G : 0.f;
                alpha = fminf(kAlphaMax, op * G);
                const float w = alpha * Tt;
                float gc = gC0 * cr + gC1 * cg + gC2 * cbl;          // <gC, colour of this Gaussian> (+ depth / alpha terms)
                if (HAS_DA) gc += gD * zd + gA;
                S -= gc * w;
                const float om = 1.f - alpha;
                const float dLda = gc * Tt - S * fast_rcp(om);
                const float dLdpow = G * (op * dLda);
                // the moments of dLdpow over the pixels (the per-record combination happens once, in the flush: k_blend_bwd2)
                const float t_my = dLdpow * dy;
                float v[10];
                v[0] = dLdpow * dx;
                v[1] = t_my;
                v[2] = v[0] * dx;
                v[3] = t_my * dx;
                v[4] = t_my * dy;
                v[5] = G * dLda;
                v[6] = w * gC0;
                v[7] = w * gC1;
                v[8] = w * gC2;
                if (HAS_DA) v[9] = w * gD;
                Tt *= om;
                // (measured: finishing the reduction with ds_add_f32 from the row leaders is 1.7x SLOWER -- four lanes on one
                //  address serialise; the transposed DPP reduction below halves the VALU cost instead)
                const float t = wave_reduce_transposed2<NV>(v, lane);
                // nine (ten) lanes of the first row hold one total each (reduce2_slot).  The row offset is a SCALAR product (j is
                // wave-uniform); left to itself the compiler folds it into a v_mad_u64_u32 per visit.
                uint32_t row;
#if defined(__HIP_DEVICE_COMPILE__)
                asm("s_mul_i32 %0, %1, %2" : "=s"(row) : "s"(j), "n"(NV));
#else
                row = (uint32_t)j * NV;
#endif
                if (part_slot >= 0) atomicAdd(&(&s_part[0][0])[part_off + row], t);
            }
        }
        __syncthreads();
        // flush: 16 lanes per Gaussian, lane r adds component r, so one atomic instruction touches 8 records of
        // 9-10 CONSECUTIVE floats (8 cache lines per wave instruction) instead of 64 scattered records -- device-scope
        // float atomics are fabric transactions on this chip, and they were 27% of this kernel when issued one
        // component at a time per lane
        {
            const int r = tid & 15, q = tid >> 4;   // 16 groups of 16 lanes
            for (int jj = q; jj < cnt; jj += NTH / 16) {
                if (r < NV) {
                    float v = s_part[jj][r];
                    if (r < 2) {        // moments -> d/d(pixel-space mean): this lane also needs the OTHER first-order moment
                        const float mo = s_part[jj][r ^ 1];
                        const float cb = s_ab[0][jj].w;                                // B'
                        const float c2 = 2.f * (r == 0 ? s_ab[0][jj].z : s_ab[1][jj].x);   // 2 A' (gx) or 2 C' (gy)
                        v = fmaf(cb, mo, c2 * v);
                    } else if (r < 5) v *= (r == 3 ? -1.f : -0.5f);                       // second moments -> conic gradients
                    __builtin_amdgcn_wave_barrier();   // every lane of the group has read both first moments before any is cleared
                    s_part[jj][r] = 0.f;   // ready for the next batch (its writers sit behind a barrier)
                    // deterministic debug mode: the (tile, instance) partial goes to its own slot, k_det_reduce sums a
                    // Gaussian's slots in list order afterwards; default: one coalesced atomic per record row
                    if (det_part) det_part[((size_t)rg.x + (size_t)b * NT + jj) * kDetStride + r] = r < 2 ? v * kLn2 : v;
                    else if (v != 0.f) atomicAdd(ggrad + (size_t)s_gid[buf][jj] * kGG + r, r < 2 ? v * kLn2 : v);
                }
            }
        }
    }
    } while (0);
    // ---- pull the next index of this list ----------------------------------------------------------------------------------
    // (at the END of the item: issued at its top -- to hide the round trip behind the item -- the returning atomic sat at the head
    //  of the wave's in-order vmcnt queue, and at t = 0 all 3 584 workgroups pull at once, ~450 per head word: the wave of thread 0
    //  could not consume its own pixel loads until its pull had come back, 220-230 us per launch against 180-190 without)
    __syncthreads();   // (s_pull: everyone has read the previous value)
    if (tid == 0) s_pull = __hip_atomic_fetch_add(&hdr->head[qx][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    qi = s_pull;
    tried = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// Optimizer-in-backward (GsrFusedAdam): gradients of one block's Gaussians sit in LDS as padded rows; the block's
// parameter / moment rows are contiguous in memory, so the update is a 16-byte stream over p, m, v with the gradient
// gathered from LDS.  `row` floats are stored per Gaussian, of which the first `nact` have a gradient in the tile
// (the rest -- SH bands above the active degree -- see g = 0 but still decay their moments, as dense Adam does).
// ------------------------------------------------------------------------------------------------
// where the next-view tail of k_preprocess_bwd<..., PREP> puts what k_preprocess would have produced for the next forward
struct PrepOut {
    CamParams cp;
    Splat* splat;
    int32_t* radii;
    uint32_t* dkey;
    uint32_t* gid;
    TileRec* rec;
    DepthHist dh;
    uint2* early_parts;      // per block: (tile instances, visible depth keys beyond the three-pass sort's window) -- see k_preprocess
    uint32_t early_window;
};

// Per-iteration densification statistics, accumulated by the per-Gaussian backward kernel (GsrDensifyStats, include/gsr.h)
struct DensDev {
    const int32_t* radii;
    float *grad_accum, *denom, *max_radii;
};

struct AdamDev {
    float* m[6];
    float* v[6];
    float step_size[6];   // lr / (1 - beta1^t)
    float b1, b2, eps;
    float inv_bc2s[6];    // 1 / sqrt(1 - beta2^t), per group (a group may be steps behind the others: GsrFusedAdam::step_lag)
    long long dp[6], dm[6], dv[6];   // byte offsets from a group's parameter / moment rows to where the UPDATED rows go (0: in place; GsrFusedAdam::param_out ...)
};
template <typename T>
__device__ __forceinline__ T* adam_out(T* p, long long d) { return reinterpret_cast<T*>(reinterpret_cast<char*>(p) + d); }

__device__ __forceinline__ void adam_rows(const float* s_g, int stride, int col0, int row, int nact, float* __restrict__ p,
                                          float* __restrict__ m, float* __restrict__ v, int nG, int tid, float step_size, float bc2,
                                          int grp, const AdamDev& ad)
{
    const int total = nG * row;
    float* const po = adam_out(p, ad.dp[grp]); float* const mo = adam_out(m, ad.dm[grp]); float* const vo = adam_out(v, ad.dv[grp]);
    if (((((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)po | (uintptr_t)mo | (uintptr_t)vo) & 15) == 0) && (total & 3) == 0) {
        float4* p4 = reinterpret_cast<float4*>(p);
        float4* m4 = reinterpret_cast<float4*>(m);
        float4* v4 = reinterpret_cast<float4*>(v);
        float4* po4 = reinterpret_cast<float4*>(po);
        float4* mo4 = reinterpret_cast<float4*>(mo);
        float4* vo4 = reinterpret_cast<float4*>(vo);
        for (int q = tid; q < total / 4; q += kPreThreads) {
            float4 pp = p4[q], mm = nt_load4(m4 + q), vv = nt_load4(v4 + q);   // moments are touched once per step: keep them out of L2
            const int f = 4 * q;
            int g = f / row, e = f - g * row;
            float gs[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                gs[c] = e < nact ? s_g[g * stride + col0 + e] : 0.f;
                if (++e == row) { e = 0; g++; }
            }
            adam_one(pp.x, gs[0], mm.x, vv.x, ad.b1, ad.b2, ad.eps, step_size, bc2);
            adam_one(pp.y, gs[1], mm.y, vv.y, ad.b1, ad.b2, ad.eps, step_size, bc2);
            adam_one(pp.z, gs[2], mm.z, vv.z, ad.b1, ad.b2, ad.eps, step_size, bc2);
            adam_one(pp.w, gs[3], mm.w, vv.w, ad.b1, ad.b2, ad.eps, step_size, bc2);
            nt_store4(po4 + q, pp); nt_store4(mo4 + q, mm); nt_store4(vo4 + q, vv);
        }
    } else {
        for (int f = tid; f < total; f += kPreThreads) {
            const int g = f / row, e = f - g * row;
            float pp = p[f], mm = m[f], vv = v[f];
            adam_one(pp, e < nact ? s_g[g * stride + col0 + e] : 0.f, mm, vv, ad.b1, ad.b2, ad.eps, step_size, bc2);
            po[f] = pp; mo[f] = mm; vo[f] = vv;
        }
    }
}

// the same update when the gradient tile is a linear copy of the rows (stage_in_lin): one 16-byte LDS read per 16-byte stream element
// KEEP: the updated parameters also replace the gradients in the tile (the next-view tail reads its Gaussian's new row there)
// (round 3, measured and dropped: keeping the rows a thread loaded for the tile in its registers -- 49 VGPRs -- and updating them
//  from there, so that the update stream does not fetch the parameter rows a second time: 311-313 us against 300-318 box to box
//  at 1 M, 305 against 261-289 for eight batched stage-A models -- the second fetch is served by L2 / the 256 MB infinity cache,
//  the registers cost more than it does)
// KP > 0: the moments of the thread's first KP stream elements were loaded at the TOP of the kernel (pm / pv), underneath the
// float64 derivative chain (round 5: the kernel is held to three waves per SIMD by its LDS tile, so ~90 VGPRs per thread sit
// idle -- they carry the loads the stream would otherwise only issue after the chain)
constexpr int kAdamPrefetch = GSR_K9_PREFETCH;
template <bool KEEP, int KP = 0>
__device__ __forceinline__ void adam_rows_lin(float* s_g, int total, float* __restrict__ p, float* __restrict__ m,
                                              float* __restrict__ v, int tid, float step_size, float bc2, int grp, const AdamDev& ad,
                                              const float4* pm = nullptr, const float4* pv = nullptr)
{
    float4* p4 = reinterpret_cast<float4*>(p);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    float4* g4 = reinterpret_cast<float4*>(s_g);
    float* const po = adam_out(p, ad.dp[grp]); float* const mo = adam_out(m, ad.dm[grp]); float* const vo = adam_out(v, ad.dv[grp]);
    float4* po4 = reinterpret_cast<float4*>(po);
    float4* mo4 = reinterpret_cast<float4*>(mo);
    float4* vo4 = reinterpret_cast<float4*>(vo);
    if ((((uintptr_t)m | (uintptr_t)v | (uintptr_t)po | (uintptr_t)mo | (uintptr_t)vo) & 15) == 0) {
        int q0 = tid;
        if constexpr (KP > 0) {
#pragma unroll
            for (int k = 0; k < KP; k++) {
                const int q = tid + k * kPreThreads;
                if (q < total / 4) {
                    float4 pp = p4[q], mm = pm[k], vv = pv[k];
                    const float4 g = g4[q];
                    adam_one(pp.x, g.x, mm.x, vv.x, ad.b1, ad.b2, ad.eps, step_size, bc2);
                    adam_one(pp.y, g.y, mm.y, vv.y, ad.b1, ad.b2, ad.eps, step_size, bc2);
                    adam_one(pp.z, g.z, mm.z, vv.z, ad.b1, ad.b2, ad.eps, step_size, bc2);
                    adam_one(pp.w, g.w, mm.w, vv.w, ad.b1, ad.b2, ad.eps, step_size, bc2);
                    nt_store4(po4 + q, pp); nt_store4(mo4 + q, mm); nt_store4(vo4 + q, vv);
                    if (KEEP) g4[q] = pp;
                }
            }
            q0 = tid + KP * kPreThreads;
        }
#pragma unroll GSR_K9_UNROLL
        for (int q = q0; q < total / 4; q += kPreThreads) {
            float4 pp = p4[q], mm = nt_load4(m4 + q), vv = nt_load4(v4 + q);
            const float4 g = g4[q];
            adam_one(pp.x, g.x, mm.x, vv.x, ad.b1, ad.b2, ad.eps, step_size, bc2);
            adam_one(pp.y, g.y, mm.y, vv.y, ad.b1, ad.b2, ad.eps, step_size, bc2);
            adam_one(pp.z, g.z, mm.z, vv.z, ad.b1, ad.b2, ad.eps, step_size, bc2);
            adam_one(pp.w, g.w, mm.w, vv.w, ad.b1, ad.b2, ad.eps, step_size, bc2);
            nt_store4(po4 + q, pp); nt_store4(mo4 + q, mm); nt_store4(vo4 + q, vv);
            if (KEEP) g4[q] = pp;
        }
    } else {
        for (int f = tid; f < total; f += kPreThreads) {
            float pp = p[f], mm = m[f], vv = v[f];
            adam_one(pp, s_g[f], mm, vv, ad.b1, ad.b2, ad.eps, step_size, bc2);
            po[f] = pp; mo[f] = mm; vo[f] = vv;
            if (KEEP) s_g[f] = pp;
        }
    }
}

// Round 5, whole blocks of the linear tile: the f_rest stream software-pipelined by hand.  On gfx9 one counter (vmcnt) counts a wave's
// loads AND stores, in order, and the in-place update makes every load of the next element a may-alias of the stores in front of it:
// the plain loop above compiles to "three loads, wait, compute, three stores, wait for nearly all of it" per element -- a load round
// trip and a store acknowledgement in series, twelve times per thread.  Here the loads of element k + D are issued BEFORE element k is
// computed and stored (different elements: no true dependence), so waiting for element k's loads never waits for a store younger than
// D elements, and D x 3 loads of 16 bytes are in flight per thread throughout.  Same arithmetic, same order per element.
#ifndef GSR_K9_DEPTH
#define GSR_K9_DEPTH 3
#endif
template <bool KEEP, int KP, int ROW, int D = GSR_K9_DEPTH>
__device__ __forceinline__ void adam_rows_lin_piped(float* s_g, float* p, float* m, float* v, int tid, float step_size, float bc2, int grp,
                                                    const AdamDev& ad, const float4* pm, const float4* pv)
{
    constexpr int total4 = kPreThreads * ROW / 4;
    constexpr int NE = (total4 + kPreThreads - 1) / kPreThreads;
    static_assert((kPreThreads * ROW) % 4 == 0, "whole 16-byte pieces");
    const float4* p4 = reinterpret_cast<const float4*>(p);
    const float4* m4 = reinterpret_cast<const float4*>(m);
    const float4* v4 = reinterpret_cast<const float4*>(v);
    float4* g4 = reinterpret_cast<float4*>(s_g);
    float4* po4 = reinterpret_cast<float4*>(adam_out(p, ad.dp[grp]));
    float4* mo4 = reinterpret_cast<float4*>(adam_out(m, ad.dm[grp]));
    float4* vo4 = reinterpret_cast<float4*>(adam_out(v, ad.dv[grp]));
    float4 P[D], M[D], V[D];
#pragma unroll
    for (int k = 0; k < D && k < NE; k++) {
        const int q = tid + k * kPreThreads;
        if ((k + 1) * kPreThreads <= total4 || q < total4) {
            P[k] = p4[q];   // (a second fetch of these rows -- the tile that held them carries the gradients now; timed with the fetch left out: 302 -> 295 us)
            if (k < KP) { M[k] = pm[k]; V[k] = pv[k]; }
            else { M[k] = nt_load4(m4 + q); V[k] = nt_load4(v4 + q); }
        }
    }
#pragma unroll
    for (int k = 0; k < NE; k++) {
        const int q = tid + k * kPreThreads, sl = k % D;
        float4 pp = P[sl], mm = M[sl], vv = V[sl];
        if (k + D < NE) {
            const int qn = q + D * kPreThreads;
            if ((k + D + 1) * kPreThreads <= total4 || qn < total4) {
                P[sl] = p4[qn];
                if (k + D < KP) { M[sl] = pm[k + D]; V[sl] = pv[k + D]; }
                else { M[sl] = nt_load4(m4 + qn); V[sl] = nt_load4(v4 + qn); }
            }
        }
        if ((k + 1) * kPreThreads <= total4 || q < total4) {
            const float4 g = g4[q];
            adam_one(pp.x, g.x, mm.x, vv.x, ad.b1, ad.b2, ad.eps, step_size, bc2);
            adam_one(pp.y, g.y, mm.y, vv.y, ad.b1, ad.b2, ad.eps, step_size, bc2);
            adam_one(pp.z, g.z, mm.z, vv.z, ad.b1, ad.b2, ad.eps, step_size, bc2);
            adam_one(pp.w, g.w, mm.w, vv.w, ad.b1, ad.b2, ad.eps, step_size, bc2);
            nt_store4(po4 + q, pp); nt_store4(mo4 + q, mm); nt_store4(vo4 + q, vv);
            if (KEEP) g4[q] = pp;
        }
    }
}

// The four small groups (mean 3, opacity 1, scaling 3, rotation 4 floats) are updated by the thread that owns the Gaussian,
// straight from its registers: neighbouring lanes touch neighbouring rows, so every fetched line is fully used, and the
// block keeps no LDS copy of these gradients (25 instead of 30.7 kB per block: one more block per CU).
// A workgroup barrier that orders LDS ONLY: `s_waitcnt lgkmcnt(0)` + `s_barrier`.  __syncthreads() carries a workgroup-scope fence, which
// on gfx9 waits for EVERY outstanding vector-memory operation of the wave (vmcnt counts loads and stores alike) -- in the optimizer-in-
// backward kernel that is a full memory round trip per barrier: the small groups' moments requested in front of the streams, the
// streams' own stores in front of the next-view tail.  What the block's threads hand each other there lives in LDS (gradient rows, the
// updated rows, the digit tables); through global memory a thread only ever reads what no thread of the kernel has written.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <bool LDS_ONLY>
__device__ __forceinline__ void k9_barrier() { if constexpr (LDS_ONLY) lds_barrier(); else __syncthreads(); }

// Round 5, the same update in two halves: the moments are REQUESTED in front of the block's row streams (adam_own_request) and the
// update is computed behind them (adam_own_apply), so the four small groups cost the kernel no memory round trip of their own -- the
// loads are in flight while the streams run, where registers are free (the parameter rows are read a second time there, from L2:
// keeping the chain's copies alive across it would cost the kernel its third wave per SIMD).
template <int K>
struct OwnMoments { float p[K], m[K], v[K]; };
template <int K>
__device__ __forceinline__ void adam_own_request(OwnMoments<K>& s, const float* p, const float* m, const float* v)
{
    if (K == 4 && ((((uintptr_t)p | (uintptr_t)m | (uintptr_t)v) & 15) == 0)) {
        const float4 pp = *reinterpret_cast<const float4*>(p);
        const float4 mm = nt_load4(reinterpret_cast<const float4*>(m)), vv = nt_load4(reinterpret_cast<const float4*>(v));
        s.p[0] = pp.x; s.p[1] = pp.y; s.p[2] = pp.z; s.p[K - 1] = pp.w;
        s.m[0] = mm.x; s.m[1] = mm.y; s.m[2] = mm.z; s.m[K - 1] = mm.w;
        s.v[0] = vv.x; s.v[1] = vv.y; s.v[2] = vv.z; s.v[K - 1] = vv.w;
        return;
    }
#pragma unroll
    for (int c = 0; c < K; c++) { s.p[c] = p[c]; s.m[c] = __builtin_nontemporal_load(m + c); s.v[c] = __builtin_nontemporal_load(v + c); }
}
template <int K>
__device__ __forceinline__ void adam_own_apply(float* p, float* m, float* v, OwnMoments<K>& s, const float* g, float step_size,
                                               float bc2, int grp, const AdamDev& ad, float* updated = nullptr)
{
    float* const po = adam_out(p, ad.dp[grp]); float* const mo = adam_out(m, ad.dm[grp]); float* const vo = adam_out(v, ad.dv[grp]);
    float pp[K];
#pragma unroll
    for (int c = 0; c < K; c++) { pp[c] = s.p[c]; adam_one(pp[c], g[c], s.m[c], s.v[c], ad.b1, ad.b2, ad.eps, step_size, bc2); }
    if (updated) {
#pragma unroll
        for (int c = 0; c < K; c++) updated[c] = pp[c];
    }
    if (K == 4 && ((((uintptr_t)po | (uintptr_t)mo | (uintptr_t)vo) & 15) == 0)) {
        nt_store4(reinterpret_cast<float4*>(po), make_float4(pp[0], pp[1], pp[2], pp[K - 1]));
        nt_store4(reinterpret_cast<float4*>(mo), make_float4(s.m[0], s.m[1], s.m[2], s.m[K - 1]));
        nt_store4(reinterpret_cast<float4*>(vo), make_float4(s.v[0], s.v[1], s.v[2], s.v[K - 1]));
        return;
    }
#pragma unroll
    for (int c = 0; c < K; c++) {
        __builtin_nontemporal_store(pp[c], po + c);
        __builtin_nontemporal_store(s.m[c], mo + c);
        __builtin_nontemporal_store(s.v[c], vo + c);
    }
}

// ------------------------------------------------------------------------------------------------
// K9: per-Gaussian backward.  SH rows in, dSH rows out through the same LDS tile (coalesced both ways).
// ------------------------------------------------------------------------------------------------
// CAM = true additionally produces dL/d(viewmatrix, projmatrix, campos) (north_star's dL/dviewmatrix; BASELINE
// config 5): per-thread contributions are reduced over the block and written as one 35-float partial per block;
// k_cam_reduce sums the partials deterministically.
#ifdef GSR_K9_TIMING   // experiment build only (tools/k9_timing.sh): where a wave of the per-Gaussian backward spends its time
constexpr int kK9DbgWaves = 16384;
__device__ unsigned long long g_k9_dbg[kK9DbgWaves * 8];   // [wave][phase], the last launch
#define K9_T(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); if ((threadIdx.x & 63) == 0) k9t[k] += now_ - k9t_last; k9t_last = now_; } while (0)
#define K9_T_FLUSH() do { const int w_ = (int)blockIdx.x * (kPreThreads / 64) + (int)(threadIdx.x >> 6); if ((threadIdx.x & 63) == 0 && w_ < kK9DbgWaves) { k9t[7] = k9t_last; for (int q_ = 0; q_ < 8; q_++) g_k9_dbg[w_ * 8 + q_] = k9t[q_]; } } while (0)
#else
#define K9_T(k) do { } while (0)
#define K9_T_FLUSH() do { } while (0)
#endif
constexpr int kCamVals = 47;   // viewmatrix 16 + projmatrix 16 + campos 3 + points_transform 12

// ADAM = true (needs RAW, shs + shs_rest, no cov_pre): optimizer-in-backward, see GsrFusedAdam in include/gsr.h.  The
// parameter pointers are then read AND written by the block that owns the rows (no __restrict__ promises on them).

// PREP >= 0 (with ADAM): "prepare in backward" -- the block also runs the NEXT render's preprocess, at SH degree PREP (the degree
// this render used, or one above it: `oneupSHdegree` between two steps, /root/reference/scene/gaussian_model_ht.py:193-195), on the
// parameters it has just updated.  -1 = off.
template <int DEG, bool RAW, bool CAM, bool ADAM, int PREP>
__global__ __launch_bounds__(kPreThreads) void k_preprocess_bwd(CamParams cp, int N, const float* means,
                                                                const float* scales, const float* rots,
                                                                const float* __restrict__ cov_pre, const float* shs,
                                                                const float* shs_rest, const float* opac_raw, AdamDev ad,
                                                                const Splat* __restrict__ splat, const float* __restrict__ ggrad,
                                                                float* __restrict__ d_means, float* __restrict__ d_means2d,
                                                                float* __restrict__ d_opac, float* __restrict__ d_colors,
                                                                float* __restrict__ d_shs, float* __restrict__ d_shs_rest,
                                                                float* __restrict__ d_scales,
                                                                float* __restrict__ d_rots, float* __restrict__ d_cov,
                                                                float* __restrict__ cam_partial, PrepOut po, DensDev ds)
{
    constexpr int NC3 = 3 * (DEG + 1) * (DEG + 1);
    __shared__ float s_sh[kPreThreads * kShStride];
    __shared__ float s_cam[CAM ? (kPreThreads / 64) * kCamVals : 1];
    // PREP (with ADAM): the updated raw parameters of this thread's Gaussian, kept for the next-view tail
    float nmean[3] = {0.f, 0.f, 0.f}, nsc[3] = {0.f, 0.f, 0.f}, nrq[4] = {1.f, 0.f, 0.f, 0.f}, nop = 0.f;
    OwnMoments<3> om_mean, om_sc;   // ADAM: the small groups' moments, requested behind the derivative chain, used behind the row streams
    OwnMoments<1> om_op;
    OwnMoments<4> om_rq;
    float own_g[11];                // their gradients: mean 3 | opacity 1 | scale 3 | rotation 4
    const int tid = threadIdx.x;
    const int base = blockIdx.x * kPreThreads;
    const int i = base + tid;
    const int nG = min(kPreThreads, N - base);
#ifdef GSR_K9_TIMING
    unsigned long long k9t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, k9t_last = __builtin_readcyclecounter();
    k9t[6] = k9t_last;   // start stamp (k9t[7] = end stamp)
#endif
    select_view(cp, (int)blockIdx.x);   // batched render: this block's model and its camera
    CamGrads cg;
    float xg[12];   // dL/d(points_transform) share of this thread
    if (CAM) {
#pragma unroll
        for (int q = 0; q < 16; q++) { cg.vm[q] = 0.f; cg.pm[q] = 0.f; }
        cg.cam[0] = cg.cam[1] = cg.cam[2] = 0.f;
#pragma unroll
        for (int q = 0; q < 12; q++) xg[q] = 0.f;
    }
    // linear tiles (see stage_in_lin): split storage of a 16-coefficient model (max_sh_degree 3, what the reference trains),
    // aligned rows.  The tile holds the FULL stored rows (3 + 45 floats per Gaussian) at every active degree: the reference
    // starts every model at active_sh_degree 0 with all 16 coefficients stored and steps the degree up once per 1 000 iterations
    // (/root/reference/scene/gaussian_model_ht.py:68,193-195; its stage-A models never leave degree 0), so bands above the active
    // degree get a zero gradient in the tile and stream through the same Adam update as the active ones (dense Adam: their
    // moments still decay).  Degree 0 reads none of the rest rows here -- the optimizer stream is their only reader.
    constexpr int NRL = 45;
    float* s_dc = s_sh;
    float* s_rest = s_sh + kPreThreads * 3;
    const bool lin = shs && shs_rest && cp.M == 16 && vec_ok(shs + (size_t)base * 3, nG * 3) &&
                     vec_ok(shs_rest + (size_t)base * NRL, nG * NRL);   // block-uniform
    // the f_rest group's moments, first kAdamPrefetch elements of this thread's share of the update stream: requested NOW
    float4 pf_m[kAdamPrefetch > 0 ? kAdamPrefetch : 1], pf_v[kAdamPrefetch > 0 ? kAdamPrefetch : 1];
    bool pf_ok = false;
    if constexpr (ADAM && kAdamPrefetch > 0) {
        if (lin && nG == kPreThreads && ad.m[2] && (((uintptr_t)ad.m[2] | (uintptr_t)ad.v[2]) & 15) == 0) {   // (block-uniform; whole blocks only)
            pf_ok = true;
            const float4* m4 = reinterpret_cast<const float4*>(ad.m[2] + (size_t)base * NRL);
            const float4* v4 = reinterpret_cast<const float4*>(ad.v[2] + (size_t)base * NRL);
#pragma unroll
            for (int k = 0; k < kAdamPrefetch; k++) { pf_m[k] = nt_load4(m4 + tid + k * kPreThreads); pf_v[k] = nt_load4(v4 + tid + k * kPreThreads); }
        }
    }
    // ADAM (round 5): what this thread reads of its OWN Gaussian -- the blend-gradient row, the raw small rows (the optimizer reads them
    // whether the Gaussian is live or not), -- is requested here, in front of the SH rows' trip through
    // LDS, so the derivative chain finds its inputs when the barrier opens instead of starting a second memory round trip there.
    float4 e_g0 = make_float4(0.f, 0.f, 0.f, 0.f), e_g1 = e_g0, e_g2 = e_g0;
    float e_mean[3] = {0.f, 0.f, 0.f}, e_sc[3] = {0.f, 0.f, 0.f}, e_rq[4] = {1.f, 0.f, 0.f, 0.f}, e_op = 0.f;
    int e_rad = 0;
    float e_ds[3] = {0.f, 0.f, 0.f}, e_m2dn = 0.f;
    if constexpr (ADAM) {
        if (i < N) {
            const float4* gp = reinterpret_cast<const float4*>(ggrad + (size_t)i * kGG);
            e_g0 = gp[0]; e_g1 = gp[1]; e_g2 = gp[2];
#pragma unroll
            for (int k = 0; k < 3; k++) { e_mean[k] = means[3 * (size_t)i + k]; e_sc[k] = scales[3 * (size_t)i + k]; }
#pragma unroll
            for (int k = 0; k < 4; k++) e_rq[k] = rots[4 * (size_t)i + k];
            e_op = opac_raw[i];
        }
    }
    if (lin) {
        // every 16-byte piece of the block's rows is requested before the first one is waited for (round 5: the copy loop
        // `tile[v] = src[v]` has a run-time trip count, hipcc does not unroll it, and each of its twelve turns was a memory round trip
        // of its own -- a fifth of a wave's life in this kernel, tools/k9_timing.sh)
        float4 r_dc[stage_regs<3>()], r_rest[stage_regs<NRL>()];
        if (nG == kPreThreads) {   // (block-uniform) a whole block: no per-piece bound but the static one of the last turn
            stage_load_full<3>(r_dc, shs + (size_t)base * 3, tid);
            if (DEG > 0) stage_load_full<NRL>(r_rest, shs_rest + (size_t)base * NRL, tid);
            stage_store_lin_full<3>(r_dc, s_dc, tid);
            if (DEG > 0) stage_store_lin_full<NRL>(r_rest, s_rest, tid);
        } else {
            stage_load<3>(r_dc, shs + (size_t)base * 3, nG, tid);
            if (DEG > 0) stage_load<NRL>(r_rest, shs_rest + (size_t)base * NRL, nG, tid);
            stage_store_lin<3>(r_dc, s_dc, nG, tid);
            if (DEG > 0) stage_store_lin<NRL>(r_rest, s_rest, nG, tid);
        }
        k9_barrier<ADAM>();
        K9_T(0);
    } else if (shs) {
        if (shs_rest) {
            const size_t row = (size_t)(cp.M - 1) * 3;
            const float* dc0 = shs + (size_t)base * 3;
            if (vec_ok(dc0, nG * 3)) stage_in_vec<3>(s_sh, 0, dc0, nG, tid);
            else for (int f = tid; f < nG * 3; f += kPreThreads) s_sh[(f / 3) * kShStride + (f % 3)] = dc0[f];
            if (NC3 > 3) {
                constexpr int NR = NC3 > 3 ? NC3 - 3 : 1;
                const float* r0 = shs_rest + (size_t)base * row;
                if (row == NR && vec_ok(r0, nG * NR)) stage_in_vec<NR>(s_sh, 3, r0, nG, tid);
                else for (int f = tid; f < nG * NR; f += kPreThreads) {
                    const int g = f / NR, e = f - g * NR;
                    s_sh[g * kShStride + 3 + e] = shs_rest[(size_t)(base + g) * row + e];
                }
            }
        } else {
            const size_t row = (size_t)cp.M * 3;
            const float* r0 = shs + (size_t)base * row;
            if (row == NC3 && vec_ok(r0, nG * NC3)) stage_in_vec<NC3>(s_sh, 0, r0, nG, tid);
            else for (int f = tid; f < nG * NC3; f += kPreThreads) {
                const int g = f / NC3, e = f - g * NC3;
                s_sh[g * kShStride + e] = shs[(size_t)(base + g) * row + e];
            }
        }
        __syncthreads();
    }
    if (i < N) {
        Camera cam = load_camera(cp);
        cam.D = DEG;
        float dmean[3] = {0.f, 0.f, 0.f}, m2d[2] = {0.f, 0.f}, dop = 0.f;
        float dsc[3] = {0.f, 0.f, 0.f}, drq[4] = {0.f, 0.f, 0.f, 0.f}, dcv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float grgb[3] = {0.f, 0.f, 0.f};
        // A Gaussian that was culled or fully occluded received nothing from the blend backward: its ggrad row is still
        // the zeros of the memset and every gradient below would come out zero -- so "row is non-zero" replaces the
        // radius > 0 test, and the 48-byte splat record is not read by this kernel at all (the activated opacity the
        // sigmoid chain needs is recomputed from the logit exactly as the forward computed it).
        const float4* gp = reinterpret_cast<const float4*>(ggrad + (size_t)i * kGG);
        const float4 g0 = ADAM ? e_g0 : gp[0], g1 = ADAM ? e_g1 : gp[1], g2 = ADAM ? e_g2 : gp[2];   // gx gy gA gB | gC gop gr gg | gb gz - -
        const bool live = g0.x != 0.f || g0.y != 0.f || g0.z != 0.f || g0.w != 0.f || g1.x != 0.f || g1.y != 0.f || g1.z != 0.f ||
                          g1.w != 0.f || g2.x != 0.f || g2.y != 0.f;
        if (live) {
            const float mraw[3] = {ADAM ? e_mean[0] : means[3 * (size_t)i], ADAM ? e_mean[1] : means[3 * (size_t)i + 1],
                                   ADAM ? e_mean[2] : means[3 * (size_t)i + 2]};
            float mean[3] = {mraw[0], mraw[1], mraw[2]};
            apply_points_transform(cp.xf, mean);
            float sc[3] = {0, 0, 0}, rq[4] = {1, 0, 0, 0}, cv[6];
            if (cov_pre) {
#pragma unroll
                for (int k = 0; k < 6; k++) cv[k] = cov_pre[6 * (size_t)i + k];
            } else {
#pragma unroll
                for (int k = 0; k < 3; k++) sc[k] = ADAM ? e_sc[k] : scales[3 * (size_t)i + k];
#pragma unroll
                for (int k = 0; k < 4; k++) rq[k] = ADAM ? e_rq[k] : rots[4 * (size_t)i + k];
            }
            float rinv = 1.f;
            if (RAW && !cov_pre) {
#pragma unroll
                for (int k = 0; k < 3; k++) sc[k] = expf(sc[k]);
                rinv = 1.0f / fmaxf(sqrtf(rq[0] * rq[0] + rq[1] * rq[1] + rq[2] * rq[2] + rq[3] * rq[3]), 1e-12f);
#pragma unroll
                for (int k = 0; k < 4; k++) rq[k] *= rinv;
            }
            GaussGrads o;
            gauss_backward(cam, mean, sc, rq, cov_pre ? cv : nullptr, g0.x, g0.y, g0.z, g0.w, g1.x, g2.y, o, CAM ? &cg : nullptr);
            dmean[0] = o.mean[0]; dmean[1] = o.mean[1]; dmean[2] = o.mean[2];
            m2d[0] = o.mean2d[0]; m2d[1] = o.mean2d[1];
            dop = g1.y;
            grgb[0] = g1.z; grgb[1] = g1.w; grgb[2] = g2.x;
            if (RAW) {   // chain through exp / normalize / sigmoid
                const float sg = 1.0f / (1.0f + expf(-(ADAM ? e_op : opac_raw[i])));   // the activated opacity, as k_preprocess computes it
                dop *= sg * (1.f - sg);
#pragma unroll
                for (int k = 0; k < 3; k++) o.scale[k] *= sc[k];
                const float dot = rq[0] * o.rot[0] + rq[1] * o.rot[1] + rq[2] * o.rot[2] + rq[3] * o.rot[3];
#pragma unroll
                for (int k = 0; k < 4; k++) o.rot[k] = (o.rot[k] - rq[k] * dot) * rinv;
            }
#pragma unroll
            for (int k = 0; k < 3; k++) dsc[k] = o.scale[k];
#pragma unroll
            for (int k = 0; k < 4; k++) drq[k] = o.rot[k];
#pragma unroll
            for (int k = 0; k < 6; k++) dcv[k] = o.cov[k];
            if (shs) {
                if (lin) sh_backward(cam, mean, s_rest + tid * NRL - 3, 3, 1, grgb, s_rest + tid * NRL - 3, 3, 1, dmean, s_dc + tid * 3, s_dc + tid * 3);
                else sh_backward(cam, mean, &s_sh[tid * kShStride], 3, 1, grgb, &s_sh[tid * kShStride], 3, 1, dmean);
                if (CAM) {   // the view direction is (p - campos)/|.|: d/dcampos = -(its share of d/dp)
                    cg.cam[0] = o.mean[0] - dmean[0]; cg.cam[1] = o.mean[1] - dmean[1]; cg.cam[2] = o.mean[2] - dmean[2];
                }
            }
            if (cp.xf) {   // chain through p' = M [p; 1]: dL/dM = dL/dp' [p; 1]^T, dL/dp = R^T dL/dp'
                if (CAM) {
#pragma unroll
                    for (int r = 0; r < 3; r++) {
                        xg[4 * r] = dmean[r] * mraw[0]; xg[4 * r + 1] = dmean[r] * mraw[1]; xg[4 * r + 2] = dmean[r] * mraw[2];
                        xg[4 * r + 3] = dmean[r];
                    }
                }
                const float d0 = dmean[0], d1 = dmean[1], d2 = dmean[2];
#pragma unroll
                for (int c = 0; c < 3; c++) dmean[c] = fmaf(cp.xf[c], d0, fmaf(cp.xf[4 + c], d1, cp.xf[8 + c] * d2));
            }
        } else if (lin) {
            for (int e = 0; e < 3; e++) s_dc[tid * 3 + e] = 0.f;
            for (int e = 0; e < NRL; e++) s_rest[tid * NRL + e] = 0.f;
        } else if (shs) {
            for (int e = 0; e < NC3; e++) s_sh[tid * kShStride + e] = 0.f;
        }
        K9_T(1);
        d_means2d[3 * (size_t)i] = m2d[0]; d_means2d[3 * (size_t)i + 1] = m2d[1]; d_means2d[3 * (size_t)i + 2] = 0.f;
        if (ADAM) {   // (the statistics' words are requested with the small groups' moments and updated behind the streams)
            if (ds.radii) { e_rad = ds.radii[i]; e_ds[0] = ds.max_radii[i]; e_ds[1] = ds.grad_accum[i]; e_ds[2] = ds.denom[i]; }
            e_m2dn = sqrtf(m2d[0] * m2d[0] + m2d[1] * m2d[1]);
        } else if (ds.radii) {   // what the reference's train_step does with radii and means2D.grad after every backward
            const int r = ds.radii[i];   // (ht3dgs_trainer.py:141-147, gaussian_model_ht.py:718-721): visible = radii > 0
            if (r > 0) {
                ds.max_radii[i] = fmaxf(ds.max_radii[i], (float)r);
                ds.grad_accum[i] += sqrtf(m2d[0] * m2d[0] + m2d[1] * m2d[1]);
                ds.denom[i] += 1.f;
            }
        }
        if (ADAM) {   // this thread is the only reader of its Gaussian's small rows, and it has read them: their moments are requested
            const size_t gi = (size_t)i;   // here and used behind the row streams (own_apply below)
            adam_own_request<3>(om_mean, means + 3 * gi, ad.m[0] + 3 * gi, ad.v[0] + 3 * gi);
            adam_own_request<1>(om_op, opac_raw + gi, ad.m[3] + gi, ad.v[3] + gi);
            adam_own_request<3>(om_sc, scales + 3 * gi, ad.m[4] + 3 * gi, ad.v[4] + 3 * gi);
            adam_own_request<4>(om_rq, rots + 4 * gi, ad.m[5] + 4 * gi, ad.v[5] + 4 * gi);
#pragma unroll
            for (int k = 0; k < 3; k++) { own_g[k] = dmean[k]; own_g[4 + k] = dsc[k]; }
            own_g[3] = dop;
#pragma unroll
            for (int k = 0; k < 4; k++) own_g[7 + k] = drq[k];
        } else {
#pragma unroll
        for (int k = 0; k < 3; k++) d_means[3 * (size_t)i + k] = dmean[k];
        d_opac[i] = dop;
        if (d_colors) {
#pragma unroll
            for (int k = 0; k < 3; k++) d_colors[3 * (size_t)i + k] = grgb[k];
        }
        if (d_scales) {
#pragma unroll
            for (int k = 0; k < 3; k++) d_scales[3 * (size_t)i + k] = dsc[k];
        }
        if (d_rots) {
#pragma unroll
            for (int k = 0; k < 4; k++) d_rots[4 * (size_t)i + k] = drq[k];
        }
        if (d_cov) {
#pragma unroll
            for (int k = 0; k < 6; k++) d_cov[6 * (size_t)i + k] = dcv[k];
        }
        }
    }
    K9_T(2);
    if (CAM) {
        const int lane = tid & 63, wave = tid >> 6;
        float cv35[kCamVals];
#pragma unroll
        for (int q = 0; q < 16; q++) { cv35[q] = cg.vm[q]; cv35[16 + q] = cg.pm[q]; }
        cv35[32] = cg.cam[0]; cv35[33] = cg.cam[1]; cv35[34] = cg.cam[2];
#pragma unroll
        for (int q = 0; q < 12; q++) cv35[35 + q] = xg[q];
#pragma unroll
        for (int q = 0; q < kCamVals; q++) {
            const float t = wave_sum_to_lane63(cv35[q]);
            if (lane == 63) s_cam[wave * kCamVals + q] = t;
        }
        __syncthreads();
        if (tid < kCamVals) {
            float t = 0.f;
            for (int w = 0; w < kPreThreads / 64; w++) t += s_cam[w * kCamVals + tid];
            cam_partial[(size_t)blockIdx.x * kCamVals + tid] = t;
        }
    }
    if (ADAM) {
        lds_barrier();   // every thread's parameters are read, every gradient row is in LDS
        const size_t b = (size_t)base;
        const int rrow = cp.M * 3 - 3;
        if (lin) {
            // whole blocks with every stream 16-byte aligned: the f_dc group's one element per thread is requested in front of the f_rest
            // pipeline and updated behind it (its loads are then the oldest in flight and its stores wait for nobody)
            const bool piped = kAdamPrefetch > 0 && pf_ok && ((ad.dp[2] | ad.dm[2] | ad.dv[2]) & 15) == 0 && (DEG > 0 || ad.m[2]) &&
                               ((((uintptr_t)(ad.m[1] + b * 3)) | ((uintptr_t)(ad.v[1] + b * 3)) | (uintptr_t)ad.dp[1] | (uintptr_t)ad.dm[1] | (uintptr_t)ad.dv[1]) & 15) == 0;
            if (piped) {
                constexpr int dc4 = kPreThreads * 3 / 4;
                float* const pd = const_cast<float*>(shs) + b * 3;
                float4 dP = make_float4(0.f, 0.f, 0.f, 0.f), dM = dP, dV = dP;
                if (tid < dc4) {
                    dP = reinterpret_cast<const float4*>(pd)[tid];
                    dM = nt_load4(reinterpret_cast<const float4*>(ad.m[1] + b * 3) + tid);
                    dV = nt_load4(reinterpret_cast<const float4*>(ad.v[1] + b * 3) + tid);
                }
                adam_rows_lin_piped<(PREP >= 0), kAdamPrefetch, NRL>(s_rest, const_cast<float*>(shs_rest) + b * NRL, ad.m[2] + b * NRL, ad.v[2] + b * NRL, tid,
                                                                     ad.step_size[2], ad.inv_bc2s[2], 2, ad, pf_m, pf_v);
                if (tid < dc4) {
                    float4* g4 = reinterpret_cast<float4*>(s_dc);
                    const float4 g = g4[tid];
                    adam_one(dP.x, g.x, dM.x, dV.x, ad.b1, ad.b2, ad.eps, ad.step_size[1], ad.inv_bc2s[1]);
                    adam_one(dP.y, g.y, dM.y, dV.y, ad.b1, ad.b2, ad.eps, ad.step_size[1], ad.inv_bc2s[1]);
                    adam_one(dP.z, g.z, dM.z, dV.z, ad.b1, ad.b2, ad.eps, ad.step_size[1], ad.inv_bc2s[1]);
                    adam_one(dP.w, g.w, dM.w, dV.w, ad.b1, ad.b2, ad.eps, ad.step_size[1], ad.inv_bc2s[1]);
                    nt_store4(reinterpret_cast<float4*>(adam_out(pd, ad.dp[1])) + tid, dP);
                    nt_store4(reinterpret_cast<float4*>(adam_out(ad.m[1] + b * 3, ad.dm[1])) + tid, dM);
                    nt_store4(reinterpret_cast<float4*>(adam_out(ad.v[1] + b * 3, ad.dv[1])) + tid, dV);
                    if (PREP >= 0) g4[tid] = dP;
                }
            } else {
            adam_rows_lin<(PREP >= 0)>(s_dc, nG * 3, const_cast<float*>(shs) + b * 3, ad.m[1] + b * 3, ad.v[1] + b * 3, tid, ad.step_size[1], ad.inv_bc2s[1], 1, ad);
            if (DEG > 0 || ad.m[2]) {   // (degree 0, moments known to be zero: the group's update is the identity -- GsrFusedAdam)
                if (kAdamPrefetch > 0 && pf_ok)
                    adam_rows_lin<(PREP >= 0), kAdamPrefetch>(s_rest, nG * NRL, const_cast<float*>(shs_rest) + b * NRL, ad.m[2] + b * NRL, ad.v[2] + b * NRL, tid,
                                                              ad.step_size[2], ad.inv_bc2s[2], 2, ad, pf_m, pf_v);
                else
                    adam_rows_lin<(PREP >= 0)>(s_rest, nG * NRL, const_cast<float*>(shs_rest) + b * NRL, ad.m[2] + b * NRL, ad.v[2] + b * NRL, tid, ad.step_size[2], ad.inv_bc2s[2], 2, ad);
            }
            }
            if (i < N) {   // the small groups' update, from the moments requested in front of the streams
                const size_t gi = (size_t)i;
                adam_own_apply<3>(const_cast<float*>(means) + 3 * gi, ad.m[0] + 3 * gi, ad.v[0] + 3 * gi, om_mean, own_g, ad.step_size[0], ad.inv_bc2s[0], 0, ad, PREP >= 0 ? nmean : nullptr);
                adam_own_apply<1>(const_cast<float*>(opac_raw) + gi, ad.m[3] + gi, ad.v[3] + gi, om_op, own_g + 3, ad.step_size[3], ad.inv_bc2s[3], 3, ad, PREP >= 0 ? &nop : nullptr);
                adam_own_apply<3>(const_cast<float*>(scales) + 3 * gi, ad.m[4] + 3 * gi, ad.v[4] + 3 * gi, om_sc, own_g + 4, ad.step_size[4], ad.inv_bc2s[4], 4, ad, PREP >= 0 ? nsc : nullptr);
                adam_own_apply<4>(const_cast<float*>(rots) + 4 * gi, ad.m[5] + 4 * gi, ad.v[5] + 4 * gi, om_rq, own_g + 7, ad.step_size[5], ad.inv_bc2s[5], 5, ad, PREP >= 0 ? nrq : nullptr);
                if (ds.radii && e_rad > 0) {   // what the reference's train_step does with radii and means2D.grad after every backward
                    ds.max_radii[i] = fmaxf(e_ds[0], (float)e_rad);   // (ht3dgs_trainer.py:141-147, gaussian_model_ht.py:718-721): visible = radii > 0
                    ds.grad_accum[i] = e_ds[1] + e_m2dn;
                    ds.denom[i] = e_ds[2] + 1.f;
                }
            }
            K9_T(3);
            if (PREP < 0) { K9_T_FLUSH(); return; }
            // ---- next-view tail ("prepare in backward", GsrNextView): this block holds the UPDATED parameters of its 128
            // Gaussians -- the small groups in the owners' registers, the SH rows in the LDS tile -- so it runs the forward
            // preprocess of the NEXT render on them right here: the next gsr_forward skips k_preprocess (no second read of the
            // 236 bytes per Gaussian, and ~2 400 VALU instructions per wave that hide under this kernel's HBM time).
            // Same functions, same order of operations as k_preprocess<PREP, true>: the records are bit-identical.
            lds_barrier();
            const bool act = i < N;
            const int model2 = select_view(po.cp, (int)blockIdx.x);
            Camera cam2 = load_camera(po.cp);
            cam2.D = PREP < 0 ? 0 : PREP;
            Splat s2;
            TileRec rec2;
            uint32_t lo2 = 0u;
            float mean2[3] = {nmean[0], nmean[1], nmean[2]};
            if (act) {
                apply_points_transform(po.cp.xf, mean2);
                float sc2[3], rq2[4] = {nrq[0], nrq[1], nrq[2], nrq[3]};
#pragma unroll
                for (int k = 0; k < 3; k++) sc2[k] = expf(nsc[k]);
                const float inv = 1.0f / fmaxf(sqrtf(rq2[0] * rq2[0] + rq2[1] * rq2[1] + rq2[2] * rq2[2] + rq2[3] * rq2[3]), 1e-12f);
#pragma unroll
                for (int k = 0; k < 4; k++) rq2[k] *= inv;
                const float op2 = 1.0f / (1.0f + expf(-nop));
                preprocess_one(cam2, mean2, sc2, rq2, nullptr, op2, nullptr, 3, 1, nullptr, s2, &rec2, true, &lo2);
            }
            count_large_rects(act, s2, rec2, po.cp.W, po.cp.H, cam2.tiles_x, cam2.tiles_y, tid);
            rec2.rect += (uint32_t)(model2 * cam2.tiles_y) << 12;   // (as k_preprocess)
            if (po.early_parts) {   // this block's share of the NEXT forward's instance count (k_preprocess, "early R")
                __shared__ uint32_t s_early2[2][kPreThreads / 64];
                const uint32_t nt = act ? s2.tiles : 0u;
                const uint32_t far = (nt > 0u && __float_as_uint(s2.depth) - kDepthKeyBias > po.early_window) ? 1u : 0u;
                const uint32_t st_ = wave_inclusive_sum(nt), sf_ = wave_inclusive_sum(far);
                if ((tid & 63) == 63) { s_early2[0][tid >> 6] = st_; s_early2[1][tid >> 6] = sf_; }
                lds_barrier();
                if (tid == 0) {
                    uint32_t t0 = 0u, t1 = 0u;
#pragma unroll
                    for (int w = 0; w < kPreThreads / 64; w++) { t0 += s_early2[0][w]; t1 += s_early2[1][w]; }
                    po.early_parts[blockIdx.x] = make_uint2(t0, t1);
                }
            }
            if (act && s2.radius > 0) {
                float col[3];
                splat_sh_color(cam2, mean2, s_rest + tid * NRL - 3, 3, 1, col, s_dc + tid * 3);
                s2.r = col[0]; s2.g = col[1]; s2.b = col[2];
            }
            uint32_t key2 = 0xffffffffu;
            if (act) {
                { Splat st = s2; st.tiles = lo2; po.splat[i] = st; }   // (as k_preprocess: the remainders in the `tiles` word)
                po.rec[i] = rec2;
                po.radii[i] = s2.radius;
                key2 = s2.tiles > 0 ? __float_as_uint(s2.depth) : 0xffffffffu;
                po.dkey[i] = key2;
                po.gid[i] = (uint32_t)i;
            }
            K9_T(4);
            if (po.dh.ghist) {
                // the depth sort's digit counts of this block's 128 keys (all in one run: 128 divides the sort's tile): four
                // 256-entry tables in the SH tile, which nobody reads any more; then one global add per non-empty bin
                lds_barrier();
                uint32_t* h = reinterpret_cast<uint32_t*>(s_sh);
#pragma unroll
                for (int q = 0; q < 1024 / kPreThreads; q++) h[q * kPreThreads + tid] = 0u;
                lds_barrier();
                if (act) {
#pragma unroll
                    for (int p = 0; p < 4; p++) atomicAdd(&h[p * 256 + ((key2 >> (8 * p)) & 255u)], 1u);
                }
                lds_barrier();
                const uint32_t x = (uint32_t)base / onesweep_run_len<uint32_t>((uint32_t)N);
#pragma unroll
                for (int q = 0; q < 1024 / kPreThreads; q++) {
                    const int e = q * kPreThreads + tid;
                    const uint32_t c = h[e];
                    if (c) atomicAdd(&po.dh.ghist[((size_t)(e >> 8) * kOsRanges + x) * 256 + (e & 255)], c);
                }
                for (uint32_t q = blockIdx.x * kPreThreads + tid; q < po.dh.status_words / 4; q += po.dh.clear_threads)
                    reinterpret_cast<uint4*>(po.dh.status)[q] = make_uint4(0u, 0u, 0u, 0u);
            }
            K9_T(5);
            K9_T_FLUSH();
            return;
        }
        if (i < N) {   // the small groups' update, from the moments requested in front of the streams
            const size_t gi = (size_t)i;
            adam_own_apply<3>(const_cast<float*>(means) + 3 * gi, ad.m[0] + 3 * gi, ad.v[0] + 3 * gi, om_mean, own_g, ad.step_size[0], ad.inv_bc2s[0], 0, ad, PREP >= 0 ? nmean : nullptr);
            adam_own_apply<1>(const_cast<float*>(opac_raw) + gi, ad.m[3] + gi, ad.v[3] + gi, om_op, own_g + 3, ad.step_size[3], ad.inv_bc2s[3], 3, ad, PREP >= 0 ? &nop : nullptr);
            adam_own_apply<3>(const_cast<float*>(scales) + 3 * gi, ad.m[4] + 3 * gi, ad.v[4] + 3 * gi, om_sc, own_g + 4, ad.step_size[4], ad.inv_bc2s[4], 4, ad, PREP >= 0 ? nsc : nullptr);
            adam_own_apply<4>(const_cast<float*>(rots) + 4 * gi, ad.m[5] + 4 * gi, ad.v[5] + 4 * gi, om_rq, own_g + 7, ad.step_size[5], ad.inv_bc2s[5], 5, ad, PREP >= 0 ? nrq : nullptr);
            if (ds.radii && e_rad > 0) {   // what the reference's train_step does with radii and means2D.grad after every backward
                ds.max_radii[i] = fmaxf(e_ds[0], (float)e_rad);   // (ht3dgs_trainer.py:141-147, gaussian_model_ht.py:718-721): visible = radii > 0
                ds.grad_accum[i] = e_ds[1] + e_m2dn;
                ds.denom[i] = e_ds[2] + 1.f;
            }
        }
        adam_rows(s_sh, kShStride, 0, 3, 3, const_cast<float*>(shs) + b * 3, ad.m[1] + b * 3, ad.v[1] + b * 3, nG, tid, ad.step_size[1], ad.inv_bc2s[1], 1, ad);
        if (DEG == 0 && !ad.m[2]) {}   // f_rest skipped: see GsrFusedAdam
        else if (rrow == 45 && NC3 == 48)
            adam_rows(s_sh, kShStride, 3, 45, 45, const_cast<float*>(shs_rest) + b * 45, ad.m[2] + b * 45, ad.v[2] + b * 45, nG, tid, ad.step_size[2], ad.inv_bc2s[2], 2, ad);
        else if (rrow > 0)
            adam_rows(s_sh, kShStride, 3, rrow, NC3 - 3, const_cast<float*>(shs_rest) + b * rrow, ad.m[2] + b * rrow, ad.v[2] + b * rrow, nG, tid,
                      ad.step_size[2], ad.inv_bc2s[2], 2, ad);
    } else if (lin && d_shs && d_shs_rest) {
        __syncthreads();
        stage_out_lin<3>(s_dc, d_shs + (size_t)base * 3, nG, tid);
        stage_out_lin<NRL>(s_rest, d_shs_rest + (size_t)base * NRL, nG, tid);
    } else if (shs && d_shs) {
        __syncthreads();
        const int row = cp.M * 3;
        // rows are contiguous in memory, so when the active degree uses every stored coefficient the store is one
        // straight stream with compile-time index arithmetic; the general case pays a run-time division
        if (d_shs_rest) {
            const int rrow = row - 3;
            float* dc0 = d_shs + (size_t)base * 3;
            if (vec_ok(dc0, nG * 3)) stage_out_vec<3>(s_sh, 0, dc0, nG, tid);
            else for (int f = tid; f < nG * 3; f += kPreThreads) dc0[f] = s_sh[(f / 3) * kShStride + (f % 3)];
            if (rrow == NC3 - 3) {
                constexpr int NR = NC3 > 3 ? NC3 - 3 : 1;
                float* r0 = d_shs_rest + (size_t)base * NR;
                if (NC3 > 3 && vec_ok(r0, nG * NR)) stage_out_vec<NR>(s_sh, 3, r0, nG, tid);
                else for (int f = tid; f < nG * NR; f += kPreThreads) r0[f] = s_sh[(f / NR) * kShStride + 3 + (f % NR)];
            } else {
                for (int f = tid; f < nG * rrow; f += kPreThreads) {
                    const int g = f / rrow, e = f - g * rrow + 3;
                    d_shs_rest[(size_t)(base + g) * rrow + (e - 3)] = e < NC3 ? s_sh[g * kShStride + e] : 0.f;
                }
            }
        } else if (row == NC3) {
            float* r0 = d_shs + (size_t)base * NC3;
            if (vec_ok(r0, nG * NC3)) stage_out_vec<NC3>(s_sh, 0, r0, nG, tid);
            else for (int f = tid; f < nG * NC3; f += kPreThreads) r0[f] = s_sh[(f / NC3) * kShStride + (f % NC3)];
        } else {
            for (int f = tid; f < nG * row; f += kPreThreads) {
                const int g = f / row, e = f - g * row;
                d_shs[(size_t)(base + g) * row + e] = e < NC3 ? s_sh[g * kShStride + e] : 0.f;
            }
        }
    }
}

// Deterministic-accumulation debug mode (gsr_set_option "deterministic_backward"): instance positions sorted by Gaussian id
// (stable: a Gaussian's instances stay in list = tile order); the first position of each run sums the run's partial
// records in that order and writes the Gaussian's row -- no float atomics, the same bits on every run.
// (round 6: the list words behind a tile's valid prefix are not written when the forward cut its lists -- ListCut -- and hold whatever
//  the buffer held before.  No wave of the backward blend visits such a position, so its slot of partials is zero; the copy of the
//  list this mode sorts only has to keep such a word INSIDE the table: an id below N adds zeros to that Gaussian's run.)
__global__ __launch_bounds__(256) void k_det_iota(uint32_t R, uint32_t* __restrict__ pos, const uint32_t* __restrict__ list, uint32_t* __restrict__ ids,
                                                  uint32_t n_gaussians)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < R) { pos[i] = i; ids[i] = min(list[i], n_gaussians - 1u); }
}
__global__ __launch_bounds__(256) void k_det_reduce(uint32_t R, const uint32_t* __restrict__ gid_sorted, const uint32_t* __restrict__ pos_sorted,
                                                    const float* __restrict__ part, float* __restrict__ ggrad)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= R) return;
    const uint32_t g = gid_sorted[i];
    if (i > 0 && gid_sorted[i - 1] == g) return;
    // (round 6: the run is summed in float64 -- the mode is a debugging aid, and with the cross-tile sum exact what is left of a
    //  gradient's error is the per-tile part: tests/test_gpu_parity.py::test_needle_gradient_error_is_the_binary32_accumulation)
    double acc[kDetStride];
#pragma unroll
    for (int r = 0; r < kDetStride; r++) acc[r] = 0.0;
    for (uint32_t j = i; j < R && gid_sorted[j] == g; j++) {
        const float* p = part + (size_t)pos_sorted[j] * kDetStride;
#pragma unroll
        for (int r = 0; r < kDetStride; r++) acc[r] += (double)p[r];
    }
#pragma unroll
    for (int r = 0; r < kDetStride; r++) ggrad[(size_t)g * kGG + r] = (float)acc[r];
}

// one block per camera entry (and, batched, per model: blockIdx.y, over that model's blocks): deterministic sum of the partials
__global__ __launch_bounds__(256) void k_cam_reduce(const float* __restrict__ partial, int nblocks, float* __restrict__ d_vm,
                                                    float* __restrict__ d_pm, float* __restrict__ d_campos, float* __restrict__ d_xf,
                                                    BatchDev bt)
{
    __shared__ double s[256];
    const int q = blockIdx.x, m = blockIdx.y;
    int lo = 0, hi = nblocks;
    if (bt.B > 1) {
#pragma unroll
        for (int k = 0; k < kMaxBatch; k++)
            if (k == m) { lo = bt.first_block[k]; hi = min(nblocks, bt.first_block[k + 1]); }
        if (d_vm) d_vm += 16 * m;
        if (d_pm) d_pm += 16 * m;
        if (d_campos) d_campos += 3 * m;
        if (d_xf) d_xf += 12 * m;
    }
    double a = 0;
    for (int b = lo + threadIdx.x; b < hi; b += 256) a += partial[(size_t)b * kCamVals + q];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float v = (float)s[0];
        if (q < 16) { if (d_vm) d_vm[q] = v; }
        else if (q < 32) { if (d_pm) d_pm[q - 16] = v; }
        else if (q < 35) { if (d_campos) d_campos[q - 32] = v; }
        else if (d_xf) d_xf[q - 35] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct GeomLayout {   // gsr_geom: only the splats survive until backward
    size_t splat;
    size_t total;
};
static GeomLayout geom_layout(int32_t N)
{
    GeomLayout g;
    g.splat = 0;
    g.total = align256((size_t)(N > 0 ? N : 1) * sizeof(Splat));
    return g;
}

struct PrepLayout {   // "prepare in backward": what k_preprocess would have produced, handed from gsr_backward to the next gsr_forward
    size_t splat, radii, dkey, gid, rec, sort, early, bytes;
};
static PrepLayout prep_layout(int32_t N)
{
    const size_t n = (size_t)(N > 0 ? N : 1);
    PrepLayout p;
    size_t o = 0;
    p.splat = o; o += align256(n * sizeof(Splat));   // == geom_layout(N): the buffer doubles as the forward's `geom`
    p.radii = o; o += align256(n * 4);
    p.dkey = o; o += align256(n * 4);
    p.gid = o; o += align256(n * 4);
    p.rec = o; o += align256(n * sizeof(TileRec));
    p.sort = o; o += align256(onesweep_scratch_bytes((uint32_t)n));   // the depth sort's scratch: digit counts filled in by the backward
    p.early = o; o += align256(((n + kPreThreads - 1) / kPreThreads) * sizeof(uint2));   // per-block shares of the next forward's R
    p.bytes = o;
    return p;
}

struct FwdScratch {   // N-sized scratch of the forward
    size_t dkey, gid, dkey_alt, gid_alt, ntiles, srec, block_sums, total, early, sort;
    size_t bytes;
};
static FwdScratch fwd_scratch_layout(int32_t N)
{
    const size_t n = (size_t)(N > 0 ? N : 1);
    FwdScratch s;
    size_t o = 0;
    s.dkey = o; o += align256(n * 4);
    s.gid = o; o += align256(n * 4);
    s.dkey_alt = o; o += align256(n * 4);
    s.gid_alt = o; o += align256(n * 4);
    s.ntiles = o; o += align256(n * sizeof(TileRec));
    s.srec = o; o += align256(n * sizeof(TileRec));
    s.block_sums = o; o += align256(((n + kEmitThreads - 1) / kEmitThreads) * 4);
    s.total = o; o += 256;
    s.early = o; o += align256(((n + kPreThreads - 1) / kPreThreads) * sizeof(uint2));   // k_preprocess's per-block shares of R (OsRider)
    s.sort = o; o += radix_scratch_bytes((uint32_t)n);
    s.bytes = o;
    return s;
}

struct BinLayout {   // persistent: list + ranges + checkpoints of long lists + the backward blend's work items
    size_t list, ranges, ckpt, items, bytes;
};

static size_t bwd_list_cap(int64_t R, size_t T) { return ((size_t)(R > 0 ? R : 0) >> 7) + T + 8; }   // sum of ceil(n_t / 128) <= R / 128 + T
static BinLayout bin_layout(int64_t R, int32_t W, int32_t H, int32_t B = 1)
{
    const size_t T = (size_t)((W + kTile - 1) / kTile) * ((H + kTile - 1) / kTile) * (size_t)(B > 1 ? B : 1);
    BinLayout b;
    b.ranges = 0;
    b.list = align256((T ? T : 1) * sizeof(uint2));
    b.ckpt = b.list + align256((size_t)(R > 0 ? R : 1) * 4);
    b.items = b.ckpt + align256((((size_t)(R > 0 ? R : 0) >> 7) + T + 1) * kCkptFloats * sizeof(float));
    // one 8-byte item per (tile, run of 128-instance batches) the backward blend has to replay: at most R / 128 + T of them
    b.bytes = b.items + kItemHdrBytes + align256(8 * bwd_list_cap(R, T) * sizeof(uint2));   // eight lists, one per XCD of the tile map
    return b;
}

struct BinScratch {
    size_t tile, tile_alt, gid_alt, sort, bytes;
};
static BinScratch bin_scratch_layout(int64_t R, int key_bytes = 4)
{
    const size_t r = (size_t)(R > 0 ? R : 1);
    BinScratch s;
    size_t o = 0;
    s.tile = o; o += align256(r * (size_t)key_bytes);
    s.tile_alt = o; o += align256(r * (size_t)key_bytes);
    s.gid_alt = o; o += align256(r * 4);
    s.sort = o; o += radix_scratch_bytes((uint32_t)r);
    s.bytes = o;
    return s;
}

// ---- speculation state, keyed per caller ----------------------------------------------------------------------------
// The capacity hint of the speculative binning is kept per (device, image size, Gaussian-count bucket): the reference
// alternates models of very different size on one process -- teacher and student (ht3dgs_trainer.py:877-883), the
// single-image models of stage A next to a leaf -- and one process-wide hint would thrash between over-allocation and
// the overflow re-run.  Buckets are half octaves of N, so a model that densifies keeps its entry.
// The pinned read-back slot and its event belong to ONE device (an event recorded on another device's stream is an
// invalid-handle error), and a slot is held by one call at a time: callers on several threads / devices do not serialise
// on each other while they enqueue or wait.
// the per-view cost cache of the balanced forward blend: one per (device, frame geometry), a handful at most
struct ViewCostCache { int dev, W, H, map, items; uint8_t* mem; size_t bytes; };
static std::vector<ViewCostCache> g_view_costs;   // guarded by g_state_mutex
static std::map<std::tuple<int, int, int, int>, bool> g_full_depth_sort;   // callers whose depths left the 27-bit window once: four 8-bit passes from then on

struct PinSlot { unsigned long long* host = nullptr; unsigned long long* dev = nullptr; hipEvent_t ev = nullptr; bool busy = false;
                 unsigned long long seq = 0;      // seq: number of the slot's last use; the scan kernel echoes it behind the count
                 // early R: what the host took from the rider, to be held against the scan's own report (words 4..6 of the slot) the next
                 // time the slot is taken or gsr_backward is entered (late_check)
                 bool pending = false; unsigned long long pending_seq = 0, pending_R = 0, pending_giveups = 0; };
static std::mutex g_state_mutex;
static std::map<int, std::vector<PinSlot*>> g_pin_slots;                     // device -> slots
static std::map<std::tuple<int, int, int, int>, uint64_t> g_hints;         // (device, W, H, bucket of N) -> capacity
static long long g_hint_override = -1;                                      // tests: capacity of the NEXT forward (one shot)
static int g_speculate = 1;
static int g_deterministic = 0;   // 1: the blend backward accumulates per Gaussian in a fixed order (debug; slower)
static int g_bwd_split = 0;  // workgroups a long tile's backward is split over (checkpoints from the forward); 1 = off; 0 = auto: 16 on a
                             // 980x545 frame, fewer the more tiles there are (the parts that find nothing to do still cost a launch slot:
                             // 16 x 17 408 workgroups for eight batched images spent 170 of 860 us on them) -- about 35 000 workgroups
static int g_ckpt_first = 1;  // 128-instance batches of a tile before the forward starts leaving checkpoints
static int g_depth_sort9 = 1;     // depth sort of large models in three 9-bit passes over (key - near-plane bits) (radix_sort.h); 0 = four 8-bit passes
static int g_tile_sort = 1;       // round 5: no global depth sort in front of the direct binning -- every tile's pairs are sorted by depth where they lie (k_tile_sort).
                                  // 0 = off (depth sort + direct binning), 1 = where it wins: tile lists of up to g_tile_sort_max_avg pairs on average (by the caller's
                                  // last R, or 4 N before there is one), 2 = wherever the direct binning runs
static int g_tile_sort_emit_max_n = 400000;   // behind the emit path (batched renders, frames above 4 096 tiles): models of up to this many Gaussians
static int g_tile_sort_max_avg = 700;   // measured (tools/ab_tile_sort*.sh, 980x545): 20 k ... 300 k Gaussians (50 ... 620 pairs per tile) -3 ... -6 % of the step,
                                        // 1 M (2 070 per tile) +6 %: the per-tile sorts move R pairs where the depth sort moves N keys
static int g_direct_bin = 1;      // tile lists by direct placement (k_chunk_counts / k_chunk_scatter) instead of emit + tile sort + ranges; 0 = the sort route
static int g_view_pose_tol_e6 = 2000;   // balanced placement without a view id: a render belongs to the cached view whose pose is within this (x 1e-6) in every matrix entry
static int g_db_slab_tiles = 0;   // frames above kDbMaxTiles tiles: tiles per slab of the slabbed scatter (0 = such frames keep the sort route)
static int g_list_cut = 1;        // round 6: tile lists written only up to where the tiles stopped at the frame's previous render (ListCut); 0 = full lists,
                                  // 1 = for models of at least g_list_cut_min_n Gaussians (where it was measured to pay), 2 = wherever the route allows
static int g_list_cut_min_n = 2000000;
static int g_list_cut_margin_e3 = 50;   // ... x (1 + this / 1000) of the depth its waves reached
static int g_list_cut_deep = 8;        // tiles whose own cut may lie behind the frame's (served by cut_chunk_tiles)
static int g_early_r = 1;         // the host learns R from the preprocess's per-block shares (published by the depth sort's histogram kernel) instead of from the scan
static int g_blend_balance = 1;   // forward blend: place the waves by the visits each took at the previous render of the same view (balance_build)
static int g_tile_map = 2;   // tile -> XCD map: 2 = 2x2 tile blocks interleaved (default), 1 = tiles interleaved, 0 = banded
static std::atomic<long long> g_spec_overflows{0}, g_spec_forwards{0}, g_exact_forwards{0}, g_depth_window_resorts{0};
// host-side time accounting of the two entry points (gsr_get_counter): wall time inside the call, and the part of the forward
// spent waiting for the instance count -- their difference is what the launching thread really works per call
static std::atomic<long long> g_fwd_calls{0}, g_fwd_ns{0}, g_fwd_wait_ns{0}, g_bwd_calls{0}, g_bwd_ns{0};
struct CallTimer {
    std::atomic<long long>&calls, &ns;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    CallTimer(std::atomic<long long>& c, std::atomic<long long>& n) : calls(c), ns(n) {}
    ~CallTimer() { calls++; ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};

static int n_bucket(int N) { return N > 0 ? (int)std::floor(2.0 * std::log2((double)N)) : 0; }

static PinSlot* acquire_pin_slot(int dev)
{
    std::lock_guard<std::mutex> lk(g_state_mutex);
    auto& v = g_pin_slots[dev];
    for (PinSlot* s : v)
        if (!s->busy) { s->busy = true; return s; }
    PinSlot* s = new PinSlot();
    if (hipHostMalloc((void**)&s->host, 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&s->dev, s->host, 0) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev, hipEventDisableTiming) != hipSuccess) { delete s; return nullptr; }
    s->host[0] = 0; s->host[1] = 0;
    for (int q = 4; q < 8; q++) s->host[q] = 0;
    s->busy = true;
    v.push_back(s);
    return s;
}
static std::atomic<long long> g_late_checks{0}, g_late_mismatches{0};
static int g_debug_late_bias = 0;
// The scan's late report of a forward that took its count early (scan_report_late).  wait: poll for it (the slot is about to be
// reused -- in a training loop the report arrived a backward ago; a back-to-back forward waits for its predecessor's scan, which runs
// long before that forward's blend ends, so the device stays fed).  Returns 0 = nothing pending / not there yet / consistent; 1 = the
// early count and the scan's total differ, or a sort look-back gave up during that forward.
static int late_check(PinSlot* s, bool wait, unsigned long long* got_total, unsigned long long* want_total)
{
    if (!s->pending) return 0;
    volatile unsigned long long* hp = s->host;
    if (hp[5] != s->pending_seq) {
        if (!wait) return 0;
        const auto limit = std::chrono::steady_clock::now() + std::chrono::milliseconds(200);
        while (hp[5] != s->pending_seq && std::chrono::steady_clock::now() < limit) cpu_relax();
        if (hp[5] != s->pending_seq) { (void)hipDeviceSynchronize(); }
        if (hp[5] != s->pending_seq) { s->pending = false; return 0; }   // (the forward never reached its scan: it failed, and said so)
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    s->pending = false;
    g_late_checks++;
    *got_total = hp[4]; *want_total = s->pending_R;
    if (hp[4] != s->pending_R || hp[6] != s->pending_giveups) { g_late_mismatches++; return 1; }
    return 0;
}
struct PinLease {   // releases the slot on every exit path
    PinSlot* s;
    explicit PinLease(PinSlot* s_) : s(s_) {}
    ~PinLease() { if (s) { std::lock_guard<std::mutex> lk(g_state_mutex); s->busy = false; } }
};

// What the forward fixed for its backward (GsrForwardOut::forward_flags): the blend kernel variant, the tile -> XCD map,
// the first checkpointed batch and whether checkpoints were written at all.  gsr_backward reads them from the flags
// instead of the process-wide options as they happen to be at backward time.
static inline int64_t pack_fwd_flags(int ppt, int tile_map, int ckpt_first)
{
    return 1 | ((int64_t)ppt << 1) | ((int64_t)tile_map << 4) | ((int64_t)ckpt_first << 6) | ((int64_t)(ppt >= 5) << 13);
}

// GsrBatch -> the kernels' BatchDev; validates the block table against N.  nullptr / B <= 1: single model.
static int batch_dev(const GsrBatch* bt, int32_t N, int32_t H, BatchDev& out)
{
    out = BatchDev{};
    out.B = 1;
    if (!bt || bt->B <= 1) return GSR_OK;
    if (bt->B > kMaxBatch || !bt->first_block) return fail(GSR_ERR_ARG, "batch: 2..16 models and a first_block table expected%s");
    const int nblocks = (N + kPreThreads - 1) / kPreThreads;
    if (bt->first_block[0] != 0 || bt->first_block[bt->B] != nblocks || (N % kPreThreads) != 0)
        return fail(GSR_ERR_ARG, "batch: first_block must run from 0 to N / 128 and N must be a multiple of 128 (pad every model)%s");
    for (int q = 0; q < bt->B; q++)
        if (bt->first_block[q + 1] < bt->first_block[q]) return fail(GSR_ERR_ARG, "batch: first_block must be non-decreasing%s");
    if ((long long)bt->B * ((H + kTile - 1) / kTile) > 4095) return fail(GSR_ERR_RANGE, "batch: more than 4095 tile rows in total%s");
    out.B = bt->B;
    for (int q = 0; q <= bt->B; q++) out.first_block[q] = bt->first_block[q];
    for (int q = bt->B + 1; q <= kMaxBatch; q++) out.first_block[q] = nblocks;
    return GSR_OK;
}

static int check_common(int32_t N, int32_t M, int32_t D, int32_t W, int32_t H)
{
    if (N < 0 || W <= 0 || H <= 0) return fail(GSR_ERR_ARG, "bad sizes%s");
    if (D < 0 || D > 3) return fail(GSR_ERR_ARG, "sh_degree must be 0..3%s");
    if (M < 0 || M > 16) return fail(GSR_ERR_ARG, "at most 16 SH coefficients per Gaussian%s");
    if (W > 4095 * kTile || H > 4095 * kTile) return fail(GSR_ERR_RANGE, "image side longer than 65520 pixels%s");
    return GSR_OK;
}

}  // namespace gsr

using namespace gsr;

extern "C" {

size_t gsr_geom_bytes(int32_t N) { return geom_layout(N).total; }
// image workspace: state planes | balanced-placement table of the forward blend (slot -> item, uint16 per workgroup, + 256 bytes of
// per-call words; single renders only) | staged counters
static size_t image_balance_offset(int32_t W, int32_t H, int32_t B) { return align256((size_t)W * H * (size_t)(B > 1 ? B : 1) * kImgPlanes * 4); }
static size_t image_balance_bytes(int32_t W, int32_t H, int32_t B)
{
    if (B > 1) return 0;
    const int tiles_x = (W + kTile - 1) / kTile, T = tiles_x * ((H + kTile - 1) / kTile);
    return 256 + align256((size_t)32 * slots_per_xcd(2, T, tiles_x) * sizeof(uint16_t) + 64);   // (map 2 has the most slots)
}
static size_t image_staged_offset(int32_t W, int32_t H, int32_t B) { return image_balance_offset(W, H, B) + image_balance_bytes(W, H, B); }
size_t gsr_image_staged_offset(int32_t W, int32_t H) { return image_staged_offset(W, H, 1); }
size_t gsr_image_bytes_batched(int32_t W, int32_t H, int32_t B)
{
    const size_t T = (size_t)((W + kTile - 1) / kTile) * ((H + kTile - 1) / kTile) * (size_t)(B > 1 ? B : 1);
    return image_staged_offset(W, H, B) + align256(T * 4 * 4);   // four per-sub-tile counters per tile
}
size_t gsr_image_bytes(int32_t W, int32_t H) { return gsr_image_bytes_batched(W, H, 1); }

int gsr_profile_read(const char* name, double* total_ms, int64_t* count)
{
    if (!name || !total_ms || !count) return GSR_ERR_ARG;
    for (int id = 0; id < P_COUNT; id++) {
        if (strcmp(name, kProfNames[id])) continue;
        std::lock_guard<std::mutex> lk(g_prof_mutex);
        double tot = 0; int64_t n = 0;
        for (auto& p : g_prof_events[id]) {
            float ms = 0.f;
            if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { tot += ms; n++; }
            (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b);
        }
        g_prof_events[id].clear();
        *total_ms = tot; *count = n;
        return GSR_OK;
    }
    return GSR_ERR_ARG;
}
size_t gsr_forward_scratch_bytes(int32_t N) { return fwd_scratch_layout(N).bytes; }
size_t gsr_binning_bytes(int64_t R, int32_t W, int32_t H) { return bin_layout(R, W, H).bytes; }
size_t gsr_binning_scratch_bytes(int64_t R) { return bin_scratch_layout(R).bytes; }
size_t gsr_backward_scratch_bytes(int32_t N)
{
    const size_t n = (size_t)(N > 0 ? N : 1);
    return align256(n * kGG * 4) + align256(((n + kPreThreads - 1) / kPreThreads) * kCamVals * 4);
}
size_t gsr_sort_scratch_bytes(uint32_t n) { return radix_scratch_bytes(n); }
size_t gsr_prepared_bytes(int32_t N) { return prep_layout(N).bytes; }
size_t gsr_prepared_radii_offset(int32_t N) { return prep_layout(N).radii; }
int gsr_prepare_supported(int32_t M, int32_t D, int32_t raw_params) { return (raw_params && M == 16 && D >= 0 && D <= 3) ? 1 : 0; }
const char* gsr_last_error(void) { return g_err; }
int gsr_version(void) { return 110; }
size_t gsr_struct_bytes(int32_t which)
{
    return which == 0 ? sizeof(GsrForwardArgs) : which == 1 ? sizeof(GsrBackwardArgs) : which == 2 ? sizeof(GsrForwardOut) : 0;
}

#ifdef GSR_K8_PHASES
int gsr_debug_k8_phases(unsigned long long* host_dst) { return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_k8ph), sizeof(unsigned long long) * 8 * 8192); }
#endif
#ifdef GSR_OS_TIMING
int gsr_debug_os_timing(unsigned long long* host_dst) { return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_os_dbg), sizeof(unsigned long long) * 8 * 4096); }
int gsr_debug_gh_timing(unsigned long long* host_dst) { return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_gh_dbg), sizeof(unsigned long long) * 8 * 512); }
#endif
#ifdef GSR_K9_TIMING
int gsr_debug_k9_timing(unsigned long long* host_dst, int reset)
{
    (void)reset;
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_k9_dbg), sizeof(unsigned long long) * 8 * kK9DbgWaves);
}
#endif
#ifdef GSR_DB_TIMING
int gsr_debug_db_timing(unsigned long long* host_dst) { return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_db_dbg), sizeof(unsigned long long) * 16); }
#endif
#ifdef GSR_K6_TIMING
int gsr_debug_k6_timing(unsigned long long* host_dst, int blocks)
{
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_k6_dbg), sizeof(unsigned long long) * 4 * (size_t)blocks);
}
int gsr_debug_k6_counts(uint32_t* host_dst, int blocks)
{
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_k6_cnt), sizeof(uint32_t) * 2 * (size_t)blocks);
}
int gsr_debug_k6_cycles(unsigned long long* host_dst, int blocks)
{
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_k6_cyc), sizeof(unsigned long long) * 2 * (size_t)blocks);
}
int gsr_debug_k8_timing(unsigned long long* host_dst, int blocks)
{
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_k8_dbg), sizeof(unsigned long long) * 4 * (size_t)blocks);
}
#endif
int gsr_set_option(const char* name, int value)
{
    if (!name) return GSR_ERR_ARG;
    // 7 (default) / 6 = one wave per 8x8 sub-tile, with / without reach bits (the tile-per-workgroup A/B kernels of rounds 1-4, values
    // 1-5, left the tree in round 5: they were compiled on request only and no default-build test reached them)
    if (!strcmp(name, "blend_fwd_ppt")) {
        if (value != 0 && value != 6 && value != 7) return GSR_ERR_ARG;
        g_blend_ppt = value; return GSR_OK;
    }
    if (!strcmp(name, "ab_variants")) return 0;   // query kept for callers of the round-4 ABI: no A/B kernels in the library
    if (!strcmp(name, "bwd_split")) { if (value < 0 || value > 64) return GSR_ERR_ARG; g_bwd_split = value; return GSR_OK; }
    if (!strcmp(name, "ckpt_first")) { if (value < 1 || value > 64) return GSR_ERR_ARG; g_ckpt_first = value; return GSR_OK; }
    if (!strcmp(name, "tile_map")) { if (value < 0 || value > 2) return GSR_ERR_ARG; g_tile_map = value; return GSR_OK; }
    if (!strcmp(name, "blend_balance")) { g_blend_balance = value ? 1 : 0; return GSR_OK; }
    if (!strcmp(name, "early_r")) { g_early_r = value ? 1 : 0; return GSR_OK; }
    if (!strcmp(name, "list_cut")) { g_list_cut = value < 0 || value > 2 ? 1 : value; return GSR_OK; }
    if (!strcmp(name, "list_cut_min_n")) { g_list_cut_min_n = value < 0 ? 2000000 : value; return GSR_OK; }
    if (!strcmp(name, "view_cache_reset")) {   // tests: every per-frame cache of the current device forgets its frames (placement costs, list cuts, counters)
        int dev_id = 0;
        if (hipGetDevice(&dev_id) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return GSR_ERR_HIP;
        std::lock_guard<std::mutex> lk(g_state_mutex);
        for (auto& c : g_view_costs)
            if (c.dev == dev_id && hipMemset(c.mem, 0, c.bytes) != hipSuccess) return GSR_ERR_HIP;
        return GSR_OK;
    }
    if (!strcmp(name, "list_cut_margin_e3")) { g_list_cut_margin_e3 = value < 0 ? 50 : value; return GSR_OK; }
    if (!strcmp(name, "list_cut_deep")) { g_list_cut_deep = value < 0 ? 8 : value; return GSR_OK; }
    if (!strcmp(name, "debug_late_bias")) { g_debug_late_bias = value; return GSR_OK; }   // tests: added ONCE to the early count a forward remembers for its late check
    if (!strcmp(name, "tile_sort")) { if (value < 0 || value > 2) return GSR_ERR_ARG; g_tile_sort = value; return GSR_OK; }
    if (!strcmp(name, "tile_sort_max_avg")) { if (value < 0) return GSR_ERR_ARG; g_tile_sort_max_avg = value; return GSR_OK; }
    if (!strcmp(name, "direct_slab_tiles")) { if (value < 0) return GSR_ERR_ARG; g_db_slab_tiles = value; return GSR_OK; }
    if (!strcmp(name, "view_pose_tol_e6")) { if (value < 0) return GSR_ERR_ARG; g_view_pose_tol_e6 = value; return GSR_OK; }
    if (!strcmp(name, "depth_sort9")) { g_depth_sort9 = value ? 1 : 0; return GSR_OK; }
    if (!strcmp(name, "direct_binning")) { g_direct_bin = value ? 1 : 0; return GSR_OK; }
    if (!strcmp(name, "speculative_binning")) { g_speculate = value ? 1 : 0; return GSR_OK; }
    if (!strcmp(name, "deterministic_backward")) { g_deterministic = value ? 1 : 0; return GSR_OK; }
    if (!strcmp(name, "binning_capacity_hint")) {   // tests: capacity of the next forward (one shot; forces an overflow re-run)
        std::lock_guard<std::mutex> lk(g_state_mutex);
        g_hint_override = value > 0 ? value : -1;
        return GSR_OK;
    }
    if (!strcmp(name, "reset_speculation")) {       // forget every capacity hint and zero the counters
        std::lock_guard<std::mutex> lk(g_state_mutex);
        g_hints.clear(); g_hint_override = -1;
        g_spec_overflows = 0; g_spec_forwards = 0; g_exact_forwards = 0; g_depth_window_resorts = 0;
        g_full_depth_sort.clear();
        return GSR_OK;
    }
    if (!strcmp(name, "profile")) { g_profile = (value == 2 || value == 3) ? value : (value ? 1 : 0); g_profile_tick = 0; return GSR_OK; }
    if (!strcmp(name, "prep_hist_max_n")) { g_prep_hist_max_n = value; return GSR_OK; }
    if (!strcmp(name, "small_sort9")) { if (value < 0) return GSR_ERR_ARG; g_small_sort9 = value; return GSR_OK; }
    if (!strcmp(name, "poll_iters")) { g_poll_iters = value < 0 ? 0 : value; return GSR_OK; }
    if (!strcmp(name, "emit_hist")) { g_emit_hist = value ? 1 : 0; return GSR_OK; }
    if (!strcmp(name, "sort_algo")) { if (value < 0 || value > 2) return GSR_ERR_ARG; g_sort_algo = value; return GSR_OK; }
    // 2 = the packed two-pixel kernel (default; 0 = default); 1 = one pixel per lane, four waves per tile (k_blend_bwd1: scenes of small splats)
    if (!strcmp(name, "blend_bwd_ppt")) {
        if (value < 0 || value > 2) return GSR_ERR_ARG;
        g_bwd_ppt = value; return GSR_OK;
    }
    return GSR_ERR_ARG;
}

// geometry + scratch of the direct binning for N Gaussians on T tiles; false: this frame keeps the sort route
struct DirectBinScratch { size_t M, GT, tbase, bsum, cut_chunk, cut_tend, cut_flag, cut_ctl, cut_bkey, cut_hist, bytes; };
static bool direct_bin_geometry(int N, int T, DirectBin& db, DirectBinScratch& ds, int tiles_x = 0, int tile_rows = 0)
{
    if (N < 1 || T < 1) return false;
    db.NS = 1; db.Ts = 0; db.Tsp = 0; db.slab_rows = 0;
    static const bool force_slabs = getenv("GSR_DB_FORCE_SLABS") != nullptr;   // (experiments: the slabbed scatter on a frame that fits one wave's tables)
    if (T > kDbMaxTiles || (force_slabs && g_db_slab_tiles > 0 && tiles_x > 0 && tile_rows > 0 && tiles_x * tile_rows == T)) {   // slabs of whole tile rows
        if (g_db_slab_tiles <= 0 || tiles_x <= 0 || tile_rows <= 0 || tiles_x * tile_rows != T || tiles_x > kDbMaxTiles || T > 32768) return false;
        const int rows = std::max(1, std::min(g_db_slab_tiles, kDbMaxTiles) / tiles_x);
        db.slab_rows = rows; db.Ts = rows * tiles_x; db.Tsp = (db.Ts + 63) & ~63; db.NS = (tile_rows + rows - 1) / rows;
        if (db.NS > 16) return false;
    }
    static std::atomic<int> cus_cached{0};   // (compute units of the first device asked about: all devices of a node are alike)
    int cus = cus_cached.load(std::memory_order_relaxed);
    if (!cus) {
        int dev = 0, n = 0;
        cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
        cus_cached.store(cus, std::memory_order_relaxed);
    }
    db.N = N; db.T = T; db.Tp = (T + 63) & ~63;
    const int lds = 12 * (db.NS > 1 ? db.Tsp : db.Tp) + 4 * kDbPairs;
    const char* env = getenv("GSR_DB_WAVES_PER_CU");   // (experiments)
    int per_cu = std::min(16, (160 * 1024) / lds);
    if (env && atoi(env) > 0) per_cu = atoi(env);
    const long long resident = std::max<long long>(1, (long long)cus * per_cu / db.NS);   // chunks such that chunks x slabs fill the chip
    const char* env_s = getenv("GSR_DB_MIN_CHUNK");
    const int min_s = env_s && atoi(env_s) > 0 ? atoi(env_s) : 128;   // (measured at 20 k - 130 k Gaussians: 128 beats 64, 256 and 512)
    long long S = ((long long)N + resident - 1) / resident;
    S = std::max<long long>(min_s, (S + 63) & ~63ll);
    if (S > 65472) return false;
    db.S = (int)S;
    db.NC = (int)(((long long)N + S - 1) / S);
    int G = 1;
    while (G * G < db.NC) G++;
    db.Cg = std::max(1, std::min((db.NC + G - 1) / G, 65535 / db.S));
    db.G = (db.NC + db.Cg - 1) / db.Cg;
    if (db.G > kDbMaxGroups) return false;
    size_t o = 0;
    ds.M = o; o += align256((size_t)db.NC * db.Tp * sizeof(uint16_t));
    ds.GT = o; o += align256((size_t)db.G * db.Tp * sizeof(uint32_t));
    ds.tbase = o; o += align256((size_t)(db.Tp + 1) * sizeof(uint32_t));
    ds.bsum = o; o += align256((size_t)db.G * ((db.Tp + 255) / 256) * sizeof(uint32_t));
    // the list cut's per-call tables (ListCut): the tiles' last chunk, valid end and flag, the control words, the chunks' first keys
    ds.cut_chunk = o; o += align256((size_t)db.Tp * sizeof(uint16_t));
    ds.cut_tend = o; o += align256((size_t)db.Tp * sizeof(uint32_t));
    ds.cut_flag = o; o += align256((size_t)db.Tp * sizeof(uint32_t));
    ds.cut_ctl = o; o += 256;
    ds.cut_bkey = o; o += align256((size_t)db.NC * sizeof(uint32_t));
    ds.cut_hist = o; o += align256((size_t)db.NC * sizeof(uint32_t));
    ds.bytes = o;
    return true;
}

int gsr_forward(const GsrForwardArgs* a, GsrForwardOut* out, void* stream_)
{
    CallTimer call_timer(g_fwd_calls, g_fwd_ns);
    hipStream_t st = (hipStream_t)stream_;
    if (!a || !out) return fail(GSR_ERR_ARG, "null args%s");
    int rc = check_common(a->N, a->M, a->D, a->W, a->H);
    if (rc) return rc;
    if (!a->out_color || !a->out_depth || !a->out_alpha || !a->image || !a->bg || !a->alloc)
        return fail(GSR_ERR_ARG, "missing output / workspace pointer%s");
    const int N = a->N, W = a->W, H = a->H;
    BatchDev bt;
    rc = batch_dev(a->batch, N, H, bt);
    if (rc) return rc;
    const int NB = bt.B;   // images rendered by this call (batched render: a tall grid of NB * tiles_y tile rows)
    // the process-wide options as they are NOW: one forward uses one consistent set and hands it to its backward
    const int opt_ppt = g_blend_ppt ? g_blend_ppt : 7, opt_map = g_tile_map, opt_ckpt = g_ckpt_first;
    const int tiles_x = (W + kTile - 1) / kTile, tiles_y = (H + kTile - 1) / kTile, T = tiles_x * tiles_y * NB;
    if (NB > 1 && opt_ppt < 6) return fail(GSR_ERR_ARG, "batch: served by the default forward blend kernel only%s");
    if (a->out_color_clamped && opt_ppt < 6) return fail(GSR_ERR_ARG, "out_color_clamped: served by the default forward blend kernel only%s");
    out->num_rendered = 0; out->binning = nullptr; out->binning_bytes = 0; out->binning_capacity = 0;
    out->forward_flags = pack_fwd_flags(opt_ppt, opt_map, opt_ckpt);
    uint64_t R = 0;
    BlendBalance bb = {};   // filled in front of k_tile_counts (whose extra workgroups build the placement the blend reads)
    ListCut lc = {};        // filled with it: the list cut lives in the same per-frame cache
    uint64_t lc_capacity = 0;   // capacity of the last launch_binning (the repair pass scatters against the same)
    Splat* splat = static_cast<Splat*>(a->geom);
    float* img = static_cast<float*>(a->image);
    uint32_t* staged = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(a->image) + image_staged_offset(W, H, NB));
    const FwdScratch L = fwd_scratch_layout(N);
    uint8_t* fs = nullptr;
    uint32_t* sorted_gid = nullptr;
    int bits = 1;
    while ((1 << bits) < T) bits++;
    const int tile_passes = (bits + 7) / 8;
    const bool wide_keys = T > 65536;   // tile ids beyond 16 bits: 32-bit keys in the instance stream
    const int nb = (N + kEmitThreads - 1) / kEmitThreads;

    // ---- R-sized state (binning result + tile-sort scratch) and the stages that need it --------------------------
    BinLayout B = bin_layout(0, W, H, NB);
    BinScratch S = bin_scratch_layout(0);
    uint8_t *bin = nullptr, *bs = nullptr;
    // direct binning (k_chunk_counts ... k_chunk_scatter) where the frame's tile tables fit a wave's LDS; else emit + tile sort
    DirectBin db = {};
    DirectBinScratch dbs = {};
    const bool direct = g_direct_bin && g_sort_algo == 2 && !wide_keys && direct_bin_geometry(N, T, db, dbs, NB == 1 ? tiles_x : 0, tiles_y);
    // tile-sort route (round 5): the binning runs over the Gaussians in index order and k_tile_sort orders every tile's pairs by depth:
    // single renders whose tile tables fit one wave (no slabs), onesweep configuration
    bool tsort = false;   // (decided below, once the caller's capacity hint is known)
    uint2* ranges = nullptr;
    uint32_t* list = nullptr;
    uint2 *pairs = nullptr, *pairs_alt = nullptr;
    auto alloc_binning = [&](uint64_t capacity) -> int {
        B = bin_layout((int64_t)capacity, W, H, NB);
        bin = static_cast<uint8_t*>(a->alloc(B.bytes, GSR_ALLOC_BINNING, a->alloc_user));
        if (!bin) return fail(GSR_ERR_ALLOC, "binning allocation failed%s");
        ranges = reinterpret_cast<uint2*>(bin + B.ranges);
        list = reinterpret_cast<uint32_t*>(bin + B.list);
        bs = nullptr;
        if (capacity > 0 && tsort) {   // the (depth key, Gaussian) pairs the scatter places, and the second buffer long segments are sorted through
            const size_t pb = align256((size_t)capacity * sizeof(uint2));
            uint8_t* pm = static_cast<uint8_t*>(a->alloc(2 * pb, GSR_ALLOC_SCRATCH, a->alloc_user));
            if (!pm) return fail(GSR_ERR_ALLOC, "pair buffer allocation failed%s");
            pairs = reinterpret_cast<uint2*>(pm);
            pairs_alt = reinterpret_cast<uint2*>(pm + pb);
        }
        if (capacity == 0 || direct) return GSR_OK;
        S = bin_scratch_layout((int64_t)capacity, wide_keys ? 4 : 2);
        bs = static_cast<uint8_t*>(a->alloc(S.bytes, GSR_ALLOC_SCRATCH, a->alloc_user));
        if (!bs) return fail(GSR_ERR_ALLOC, "binning scratch allocation failed%s");
        return GSR_OK;
    };
    const TileRec* ts_rec = nullptr;      // (tile-sort route) the tile records and depth keys where the preprocess left them
    const uint32_t* ts_dkey = nullptr;
    // emit + tile sort + ranges for `capacity` instances; n_dev != nullptr: the real count is read on the device.
    // prezeroed: ranges, the staged counters and the head of the sort scratch were cleared by k_block_scan.
    auto launch_binning_t = [&](auto key_tag, uint64_t capacity, const unsigned long long* n_dev, bool prezeroed) -> int {
        using KeyT = decltype(key_tag);
        if (!prezeroed) GSR_HIP(hipMemsetAsync(ranges, 0, (size_t)T * sizeof(uint2), st));
        if (capacity == 0) return GSR_OK;
        KeyT* tkey = reinterpret_cast<KeyT*>(bs + S.tile);
        KeyT* tkey_alt = reinterpret_cast<KeyT*>(bs + S.tile_alt);
        uint32_t* gid_alt2 = reinterpret_cast<uint32_t*>(bs + S.gid_alt);
        uint32_t* block_sums = reinterpret_cast<uint32_t*>(fs + L.block_sums);
        // arrange the ping-pong so that the sorted gids land directly in `list`
        uint32_t* v0 = (tile_passes & 1) ? gid_alt2 : list;
        uint32_t* v1 = (tile_passes & 1) ? list : gid_alt2;
        // the emission counts the tile sort's digits itself when that sort's head is known to be zero (cleared by k_block_scan)
        const bool emit_hist = g_sort_algo == 2 && prezeroed && g_emit_hist;
        EmitHist eh = {};
        if (emit_hist) {
            eh.ghist = reinterpret_cast<uint32_t*>(bs + S.sort);
            eh.status = onesweep_status(bs + S.sort);
            eh.status_words = onesweep_status_words<KeyT>((uint32_t)capacity, bits);
            eh.bits = bits;
            eh.n_dev = n_dev;
        }
        {
            ProfScope ps(P_EMIT, st);
            hipLaunchKernelGGL(k_emit<KeyT>, dim3(nb), dim3(kEmitThreads), 0, st, N, W, H, tiles_x, tiles_y, sorted_gid, splat,
                               tsort ? ts_rec : reinterpret_cast<const TileRec*>(fs + L.srec), block_sums, tkey, v0,
                               (uint32_t)capacity, eh);
        }
        int in_alt = 0;
        {
            ProfScope ps(P_SORT_TILE, st);
            GSR_HIP(g_sort_algo == 2 ? onesweep_sort_pairs<KeyT>(tkey, v0, tkey_alt, v1, (uint32_t)capacity, 0, bits, bs + S.sort,
                                                                 &in_alt, st, n_dev, prezeroed, emit_hist)
                                     : radix_sort_pairs<KeyT>(tkey, v0, tkey_alt, v1, (uint32_t)capacity, 0, tile_passes * 8, bs + S.sort,
                                                              &in_alt, st));
        }
        const KeyT* skey = in_alt ? tkey_alt : tkey;
        {
            ProfScope ps(P_RANGES, st);
            constexpr uint32_t kpb = 256u * (16u / (uint32_t)sizeof(KeyT));   // keys per block
            hipLaunchKernelGGL(k_tile_ranges<KeyT>, dim3(((uint32_t)capacity + kpb - 1) / kpb), dim3(256), 0, st, (uint32_t)capacity, skey, ranges, n_dev);
        }
        if (tsort) {   // the lists came out in index order (no depth sort in front, the tile sort is stable): every tile's by depth now
            ProfScope ps(P_SORT_DEPTH, st);
            hipLaunchKernelGGL(k_tile_sort_wave<true>, dim3(T), dim3(64), 0, st, ranges, pairs, list, T, ts_dkey);
            hipLaunchKernelGGL((k_tile_sort<1024, true>), dim3(std::min(T, 512)), dim3(1024), 0, st, ranges, pairs, pairs_alt, list, T, ts_dkey);
        }
        return GSR_OK;
    };
    auto launch_binning = [&](uint64_t capacity, const unsigned long long* n_dev, bool prezeroed) -> int {
        if (direct) {
            if (capacity == 0) { GSR_HIP(hipMemsetAsync(ranges, 0, (size_t)T * sizeof(uint2), st)); return GSR_OK; }
            const int per = (db.NC + 7) / 8, grid = std::max(8 * per * db.NS, (T + 63) / 64);
            if (tsort) {
                {
                    ProfScope ps(P_EMIT, st);
                    hipLaunchKernelGGL((k_chunk_scatter<false, true>), dim3(grid), dim3(64), (size_t)12 * db.Tp + 4 * kDbPairs, st, db, W, H, tiles_x, tiles_y,
                                       (const uint32_t*)nullptr, ts_rec, splat, list, ranges, (uint32_t)std::min<uint64_t>(capacity, 0xffffffffull), ts_dkey, pairs);
                }
                ProfScope ps2(P_SORT_TILE, st);
                hipLaunchKernelGGL(k_tile_sort_wave<false>, dim3(T), dim3(64), 0, st, ranges, pairs, list, T, (const uint32_t*)nullptr);
                hipLaunchKernelGGL(k_tile_sort<1024>, dim3(std::min(T, 512)), dim3(1024), 0, st, ranges, pairs, pairs_alt, list, T);
                return GSR_OK;
            }
            ProfScope ps(P_EMIT, st);
            if (db.NS > 1)
                hipLaunchKernelGGL(k_chunk_scatter<true>, dim3(grid), dim3(64), (size_t)12 * db.Tsp + 4 * kDbPairs, st, db, W, H, tiles_x, tiles_y, sorted_gid,
                                   reinterpret_cast<const TileRec*>(fs + L.srec), splat, list, ranges, (uint32_t)std::min<uint64_t>(capacity, 0xffffffffull));
            else if (lc.key) {
                lc_capacity = capacity;
                hipLaunchKernelGGL((k_chunk_scatter<false, false, 1>), dim3(grid), dim3(64), (size_t)12 * db.Tp + 4 * kDbPairs, st, db, W, H,
                                   tiles_x, tiles_y, sorted_gid, reinterpret_cast<const TileRec*>(fs + L.srec), splat, list, ranges,
                                   (uint32_t)std::min<uint64_t>(capacity, 0xffffffffull), (const uint32_t*)nullptr, (uint2*)nullptr, lc);
            } else
                hipLaunchKernelGGL(k_chunk_scatter<false>, dim3(grid), dim3(64), (size_t)12 * db.Tp + 4 * kDbPairs, st, db, W, H, tiles_x, tiles_y, sorted_gid,
                                   reinterpret_cast<const TileRec*>(fs + L.srec), splat, list, ranges, (uint32_t)std::min<uint64_t>(capacity, 0xffffffffull));
            return GSR_OK;
        }
        return wide_keys ? launch_binning_t(uint32_t{}, capacity, n_dev, prezeroed) : launch_binning_t(uint16_t{}, capacity, n_dev, prezeroed);
    };
    auto launch_blend = [&](bool prezeroed) -> int {
        const int ppt = opt_ppt;   // default 7: one wave per 8x8 sub-tile, sign-encoded done + sub-tile reach bits (6: without the bits, 5: lane mask)
        if (!prezeroed) GSR_HIP(hipMemsetAsync(staged, 0, (size_t)T * 16, st));
        float* ckpt = reinterpret_cast<float*>(bin + B.ckpt);
        if (ppt == 7 && g_profile && (g_profile != 3 || g_profile_tick.fetch_add(1u) % 3u == 0u)) {
            // timed launch of the default kernel (bench.py's in-run roofline timing): the dispatch's OWN start / stop timestamps
            // (hipExtLaunchKernelGGL) instead of an event record in front of and behind it -- each of those is a barrier packet that
            // idles the queue for ~6 us (tools/api_timeline.sh: 12 us per step of the timed region went to the measurement)
            hipEvent_t ea = nullptr, eb = nullptr;
            if (hipEventCreate(&ea) != hipSuccess || hipEventCreate(&eb) != hipSuccess) return fail(GSR_ERR_HIP, "hipEventCreate failed%s");
            hipExtLaunchKernelGGL(k_blend_fwd_w6<true>, dim3(8 * 4 * slots_per_xcd(opt_map, T, tiles_x)), dim3(64), 0, st, ea, eb, 0, W, H, tiles_x, T, ranges, list,
                                  splat, a->bg, a->out_color, a->out_depth, a->out_alpha, img, staged, opt_map, ckpt, opt_ckpt, tiles_y, bb, a->out_color_clamped,
                                  lc, lc.key ? 1 : 0);
            std::lock_guard<std::mutex> lk(g_prof_mutex);
            g_prof_events[P_BLEND_FWD].push_back({ea, eb});
        } else {
            ProfScope ps(ppt == 7 && g_profile == 3 ? P_COUNT : P_BLEND_FWD, st);   // (mode 3, an untimed launch: no events)
            if (ppt == 7)
                hipLaunchKernelGGL(k_blend_fwd_w6<true>, dim3(8 * 4 * slots_per_xcd(opt_map, T, tiles_x)), dim3(64), 0, st, W, H, tiles_x, T, ranges, list, splat, a->bg,
                                   a->out_color, a->out_depth, a->out_alpha, img, staged, opt_map, ckpt, opt_ckpt, tiles_y, bb, a->out_color_clamped,
                                   lc, lc.key ? 1 : 0);
            else if (ppt == 6)
                hipLaunchKernelGGL(k_blend_fwd_w6<false>, dim3(8 * 4 * slots_per_xcd(opt_map, T, tiles_x)), dim3(64), 0, st, W, H, tiles_x, T, ranges, list, splat, a->bg,
                                   a->out_color, a->out_depth, a->out_alpha, img, staged, opt_map, ckpt, opt_ckpt, tiles_y, BlendBalance{}, a->out_color_clamped);
            else return fail(GSR_ERR_ARG, "unknown forward blend variant%s");
        }
        if (ppt == 7 && lc.key && lc_capacity > 0) {
            // the repair pass of the list cut: the same scatter over the chunks behind the flagged tiles' cuts and the same blend over
            // the flagged tiles' full lists -- every workgroup of both launches reads one word and leaves unless a wave flagged a tile
            ProfScope ps(P_CUT_REPAIR, st);
            const int per = (db.NC + 7) / 8, grid = std::max(8 * per * db.NS, (T + 63) / 64);
            hipLaunchKernelGGL((k_chunk_scatter<false, false, 2>), dim3(grid), dim3(64), (size_t)12 * db.Tp + 4 * kDbPairs, st, db, W, H,
                               tiles_x, tiles_y, sorted_gid, reinterpret_cast<const TileRec*>(fs + L.srec), splat, list, ranges,
                               (uint32_t)std::min<uint64_t>(lc_capacity, 0xffffffffull), (const uint32_t*)nullptr, (uint2*)nullptr, lc);
            hipLaunchKernelGGL(k_blend_fwd_w6<true>, dim3(8 * 4 * slots_per_xcd(opt_map, T, tiles_x)), dim3(64), 0, st, W, H, tiles_x, T, ranges, list, splat, a->bg,
                               a->out_color, a->out_depth, a->out_alpha, img, staged, opt_map, ckpt, opt_ckpt, tiles_y, bb, a->out_color_clamped, lc, 2);
        }
        GSR_HIP(hipGetLastError());
        return GSR_OK;
    };

    if (N == 0) {
        rc = alloc_binning(0);
        if (rc) return rc;
        rc = launch_binning(0, nullptr, false);
        if (rc) return rc;
        rc = launch_blend(false);
        if (rc) return rc;
        out->binning = bin;
        out->binning_bytes = B.bytes;
        return GSR_OK;
    }

    if (!a->means3D || !a->opacities || !a->geom || !a->radii || !a->viewmatrix || !a->projmatrix)
        return fail(GSR_ERR_ARG, "missing input pointer%s");
    if ((a->shs == nullptr) == (a->colors_precomp == nullptr)) return fail(GSR_ERR_ARG, "provide exactly one of shs / colors_precomp%s");
    if ((a->cov3D_precomp == nullptr) == (a->scales == nullptr || a->rotations == nullptr))
        return fail(GSR_ERR_ARG, "provide exactly one of (scales, rotations) / cov3D_precomp%s");
    if (a->shs && (!a->campos || a->M < (a->D + 1) * (a->D + 1))) return fail(GSR_ERR_ARG, "shs needs campos and M >= (D+1)^2%s");

    // ---- the instance count R only exists on the device.  Classic flow: read it back, synchronise, size the
    // R-dependent buffers and grids exactly.  That puts a host round trip (~35 us of idle GPU per frame) in the middle
    // of every forward, so by default the binning is launched SPECULATIVELY against a capacity (1.25x the recent
    // frames' R): kernels take the true count from device memory, the read-back is only waited for after everything
    // is enqueued, and in the rare overflow (R > capacity) the binning is simply launched again with the exact size.
    // Knowing the capacity up front also lets the small clears (ranges, staged counters, sort-scratch heads) ride in
    // kernels that run anyway instead of five separate fill launches. ----
    int dev_id = 0;
    GSR_HIP(hipGetDevice(&dev_id));
    const auto hint_key = std::make_tuple(dev_id, W, H * NB, n_bucket(N));
    uint64_t hint = 0;
    {
        std::lock_guard<std::mutex> lk(g_state_mutex);
        if (g_hint_override > 0) { hint = (uint64_t)g_hint_override; g_hint_override = -1; }
        else { auto it = g_hints.find(hint_key); if (it != g_hints.end()) hint = it->second; }
    }
    const bool speculative = g_speculate && g_sort_algo == 2 && hint > 0;
    const uint64_t cap = speculative ? hint : 0;
    {
        const int ts_mode = g_tile_sort;
        const uint64_t avg = (hint > 0 ? hint : (uint64_t)N * 4u) / (uint64_t)T;
        // (behind the direct binning: single renders whose tile tables fit one wave; behind the emit path -- batched renders, frames above
        //  4 096 tiles -- wherever both sorts are the onesweep ones)
        // (on the emit path the keys are gathered per pair and the lists of a large model spread widely: measured at 1920x1080, 300 k
        //  Gaussians -1.7 % of the step, 1 M +1 ... +12 %; eight stage-A models in one launch chain -2.5 %: small models only)
        const bool auto_ok = ts_mode == 1 && avg <= (uint64_t)g_tile_sort_max_avg && (direct || N / std::max(1, NB) <= g_tile_sort_emit_max_n);
        tsort = (ts_mode == 2 || auto_ok) && g_sort_algo == 2 && (direct ? (NB == 1 && db.NS == 1) : true);
    }

    fs = static_cast<uint8_t*>(a->alloc(align256(L.bytes) + (direct ? dbs.bytes : 0), GSR_ALLOC_SCRATCH, a->alloc_user));   // (+ the chunk tables of the direct binning)
    if (!fs) return fail(GSR_ERR_ALLOC, "scratch allocation failed%s");
    if (speculative) {
        rc = alloc_binning(cap);
        if (rc) return rc;
    }
    uint32_t* dkey = reinterpret_cast<uint32_t*>(fs + L.dkey);
    uint32_t* gid = reinterpret_cast<uint32_t*>(fs + L.gid);
    uint32_t* dkey_alt = reinterpret_cast<uint32_t*>(fs + L.dkey_alt);
    uint32_t* gid_alt = reinterpret_cast<uint32_t*>(fs + L.gid_alt);
    TileRec* ntiles = reinterpret_cast<TileRec*>(fs + L.ntiles);
    uint32_t* block_sums = reinterpret_cast<uint32_t*>(fs + L.block_sums);
    unsigned long long* total = reinterpret_cast<unsigned long long*>(fs + L.total);
    const bool depth_onesweep = g_sort_algo != 0;
    CamParams cp = {a->viewmatrix, a->projmatrix, a->campos, a->tanfovx, a->tanfovy, a->scale_modifier, W, H, a->D, a->M, a->points_transform, bt};
    const int grid = (N + kPreThreads - 1) / kPreThreads;
    // block 0 of k_preprocess clears the head (digit histograms + tickets) of the depth sort's scratch
    uint32_t* zero_words = depth_onesweep ? reinterpret_cast<uint32_t*>(fs + L.sort) : nullptr;
    const int zero_count = depth_onesweep ? (int)kOnesweepHeadWordsMax : 0;   // (either layout of the depth sort's head)
    uint8_t* depth_scratch = fs + L.sort;
    bool depth_hist_done = false;
    // early R (round 5): a speculative forward that runs its own preprocess learns R from k_preprocess's per-block shares, summed and
    // published by the depth sort's histogram kernel (radix_sort.h OsRider) -- the host's wait for R at the end of this function then
    // ends ~10 us into the forward instead of behind sort + counts + scan, and the caller can enqueue its loss and its backward while
    // the forward is still running.  (A forward that takes a prepared buffer finds the shares in it: the backward that ran its
    // preprocess left them there, k_preprocess_bwd's next-view tail.)
    const bool early_r = g_early_r && speculative && depth_onesweep;
    uint2* early_parts = early_r ? reinterpret_cast<uint2*>(a->prepared ? static_cast<uint8_t*>(a->prepared) + prep_layout(N).early : fs + L.early) : nullptr;
    // visible depth keys beyond the 27-bit window of the three-pass sort are counted whenever this forward MAY sort on it
    const uint32_t early_window = (N > g_prep_hist_max_n || (g_small_sort9 && !a->prepared && N >= g_small_sort9)) ? (1u << 27) - 1u : 0xffffffffu;
    if (a->prepared) {
        // "prepare in backward": the preceding gsr_backward already ran the preprocess of this render on the updated
        // parameters (k_preprocess_bwd<..., PREP>); its records, sort keys and tile records are taken from the hand-over buffer
        if (a->prepared != a->geom) return fail(GSR_ERR_ARG, "prepared: geom must be the prepared buffer itself%s");
        const PrepLayout PL = prep_layout(N);
        uint8_t* pb = static_cast<uint8_t*>(a->prepared);
        dkey = reinterpret_cast<uint32_t*>(pb + PL.dkey);
        gid = reinterpret_cast<uint32_t*>(pb + PL.gid);
        ntiles = reinterpret_cast<TileRec*>(pb + PL.rec);
        ProfScope ps(P_PRE_FWD, st);
        // the depth sort's scratch is the buffer's too: its head was cleared in the backward (no memset here), and for small
        // models the backward also counted the digits and cleared the status words, so the sort starts with its first pass
        // (onesweep_sort_pairs hist_done)
        if (depth_onesweep) { depth_scratch = pb + PL.sort; depth_hist_done = prep_counts_digits(N); }
        // radii: a caller that takes them straight from the buffer (gsr_prepared_radii_offset) passes its own pointer into it
        // and nothing at all runs in front of the sort; otherwise a 4 N-byte copy kernel
        if (a->radii != reinterpret_cast<const int32_t*>(pb + PL.radii))
            hipLaunchKernelGGL(k_prepared_begin, dim3((N + 255) / 256), dim3(256), 0, st, N, reinterpret_cast<const int32_t*>(pb + PL.radii),
                               a->radii, (uint32_t*)nullptr, 0);
    } else {
#define GSR_PRE_(DEG, RAW)                                                                                                          \
    hipLaunchKernelGGL((k_preprocess<DEG, RAW>), dim3(grid), dim3(kPreThreads), 0, st, cp, N, a->means3D, a->scales, a->rotations, \
                       a->cov3D_precomp, a->opacities, a->shs, a->shs_rest, a->colors_precomp, splat, a->radii, dkey, gid, ntiles,  \
                       zero_words, zero_count, 0, DepthHist{}, early_parts, early_window, a->visible)
#define GSR_PRE(DEG) do { if (a->raw_params) GSR_PRE_(DEG, true); else GSR_PRE_(DEG, false); } while (0)
    {
        ProfScope ps(P_PRE_FWD, st);
        switch (a->shs ? a->D : 0) {
            case 0: GSR_PRE(0); break;
            case 1: GSR_PRE(1); break;
            case 2: GSR_PRE(2); break;
            default: GSR_PRE(3); break;
        }
    }
#undef GSR_PRE
#undef GSR_PRE_
    }
    int in_alt = 0;
    // large models: three 9-bit passes over the 27-bit depth window (radix_sort.h); small ones keep the four 8-bit passes whose
    // digits the producer of the keys has counted (no histogram launch).  A caller whose depths left the window once stays on
    // the full sort.
    // (round 5: a forward that runs its own preprocess launches a digit histogram anyway, so below the threshold too three 9-bit passes
    //  behind it beat four 8-bit ones -- one latency-bound pass less; "small_sort9" = the smallest model that takes them, 0 = off)
    bool wide_depth = depth_onesweep && g_depth_sort9 && !depth_hist_done && (N > g_prep_hist_max_n || (g_small_sort9 && N >= g_small_sort9));
    if (wide_depth) {
        std::lock_guard<std::mutex> lk(g_state_mutex);
        auto it = g_full_depth_sort.find(hint_key);
        if (it != g_full_depth_sort.end() && it->second) wide_depth = false;
    }
    const unsigned int* window_overflow = wide_depth ? onesweep_overflow_word(depth_scratch) : nullptr;   // (in the sort's zeroed scratch head)
    if (tsort) wide_depth = false;   // (no depth sort, no window)
    const unsigned int* window_overflow_ = tsort ? nullptr : window_overflow;
    PinLease pin(acquire_pin_slot(dev_id));
    if (!pin.s) return fail(GSR_ERR_HIP, "pinned read-back slot allocation failed%s");
    {   // the slot's previous forward took its count early: hold it against what that forward's scan reported (scan_report_late)
        unsigned long long got = 0, want = 0;
        if (late_check(pin.s, true, &got, &want)) {
            char msg[200];
            snprintf(msg, sizeof msg, "a PREVIOUS forward's binning was invalid: early instance count %llu, its scan counted %llu (or a sort look-back gave up)%%s", want, got);
            return fail(GSR_ERR_HIP, msg);
        }
    }
    OsRider rider = {};
    if (early_r) { rider.parts = early_parts; rider.nparts = (uint32_t)grid; rider.host = pin.s->dev; rider.seq = ++pin.s->seq; }
    ts_rec = ntiles; ts_dkey = dkey;
    if (!tsort) {
        ProfScope ps(P_SORT_DEPTH, st);
        GSR_HIP(depth_onesweep ? onesweep_sort_pairs<uint32_t>(dkey, gid, dkey_alt, gid_alt, (uint32_t)N, 0, wide_depth ? 27 : 32, depth_scratch, &in_alt, st,
                                                               nullptr, true, depth_hist_done, wide_depth ? 9 : 8, kDepthKeyBias, early_r ? &rider : nullptr)
                               : radix_sort_pairs<uint32_t>(dkey, gid, dkey_alt, gid_alt, (uint32_t)N, 0, 32, fs + L.sort, &in_alt, st));
    }
    sorted_gid = tsort ? nullptr : (in_alt ? gid_alt : gid);
    // tiles-touched in depth order: R to the pinned slot, and what the binning needs (block offsets / the chunk tables)
    auto launch_counts = [&](const BlendBalance& bal, const ZeroJobs& zjobs, const unsigned int* wo, unsigned long long seq, bool publish = true) {
        TileRec* srec = reinterpret_cast<TileRec*>(fs + L.srec);
        unsigned long long* const host_slot = publish ? pin.s->dev : nullptr;   // (early R: the histogram kernel already told the host)
        unsigned long long* const late_slot = publish ? nullptr : pin.s->dev + 4;  // (... and the scan reports behind it: scan_report_late)
        if (direct) {
            // (tile-sort route: index order, no sorted copy of the records, and -- no depth sort to ride on -- the rider that publishes R)
            hipLaunchKernelGGL(k_chunk_counts, dim3(db.NC + (bal.hdr ? 8 : 0)), dim3(kEmitThreads), (size_t)2 * db.Tp, st, db, W, H,
                               tiles_x, tiles_y, sorted_gid, ntiles, splat, tsort ? (TileRec*)nullptr : srec, bal,
                               (tsort && early_r && !publish) ? rider : OsRider{}, lc);
            hipLaunchKernelGGL(k_chunk_scan1, dim3((db.Tp + 255) / 256, db.G), dim3(256), 0, st, db);
            hipLaunchKernelGGL(k_chunk_scan2, dim3((db.Tp + 255) / 256), dim3(256), lc.key ? (size_t)db.NC * 4 : 0, st, db, total, zjobs, host_slot, seq, wo,
                               late_slot, lc);
        } else {
            hipLaunchKernelGGL(k_tile_counts, dim3(nb + (bal.hdr ? 8 : 0)), dim3(kEmitThreads), 0, st, N, sorted_gid, ntiles, block_sums,
                               tsort ? (TileRec*)nullptr : srec, bal, nb, (tsort && early_r && !publish) ? rider : OsRider{});
            // the scan writes R straight into the pinned slot (device-visible host memory): no copy launch behind it
            hipLaunchKernelGGL(k_block_scan, dim3(1), dim3(1024), 0, st, block_sums, nb, total, zjobs, host_slot, seq, wo, late_slot);
        }
    };
    {
        ProfScope ps(P_SCAN, st);
        ZeroJobs zj = {};
        if (speculative) {   // the single-workgroup scan also clears what the binning / blend expect to be zero
            zj.p[0] = ranges; zj.words[0] = (uint32_t)((size_t)T * sizeof(uint2) / 4);
            zj.p[1] = staged; zj.words[1] = (uint32_t)T * 4u;
            if (bs) { zj.p[2] = bs + S.sort; zj.words[2] = kOnesweepHeadWords; }
        }
        // balanced placement of the forward blend (balance_build): single renders through the default kernel, frames of up to
        // 65 535 sub-tile waves; the per-view cost cache is a lazily allocated device buffer per (device, frame geometry)
        {
            const int nslots4 = 4 * slots_per_xcd(opt_map, T, tiles_x), items = 8 * nslots4;
            if (g_blend_balance && NB == 1 && opt_ppt == 7 && items <= 65535 && nslots4 <= 20 * kEmitThreads && a->viewmatrix) {
                const size_t hdr_bytes = align256(sizeof(ViewCostHdr));
                // header | visit counts [entries][items] | list cut: depth keys [entries][T], the model size they belong to [entries]
                const size_t vc_cut = align256(hdr_bytes + (size_t)kVcEntries * items * sizeof(uint16_t));
                const size_t vc_own = vc_cut + align256((size_t)kVcEntries * T * sizeof(uint32_t));
                const size_t vc_bytes = vc_own + align256((size_t)kVcEntries * sizeof(uint32_t));
                uint8_t* mem = nullptr;
                {
                    std::lock_guard<std::mutex> lk(g_state_mutex);
                    for (auto& c : g_view_costs)
                        if (c.dev == dev_id && c.W == W && c.H == H && c.map == opt_map && c.items == items) mem = c.mem;
                    if (!mem && g_view_costs.size() < 64) {   // (frame geometries per process; round 6: 64 -- a test session renders more than 16 sizes)
                        if (hipMalloc((void**)&mem, vc_bytes) == hipSuccess) {
                            // cleared SYNCHRONOUSLY, once per (device, frame geometry), before anybody can look at it (with
                            // pose-keyed entries an uncleared header is not "garbage hashes = a miss" -- a key that reads 2 with pose
                            // floats inside the tolerance would be a hit on garbage cost tables); a failed clear = no cache
                            if (hipMemset(mem, 0, vc_bytes) == hipSuccess) {
                                g_view_costs.push_back({dev_id, W, H, opt_map, items, mem, vc_bytes});
                            } else {
                                (void)hipGetLastError();
                                (void)hipFree(mem);
                                mem = nullptr;
                            }
                        } else {
                            (void)hipGetLastError();
                            mem = nullptr;
                        }
                    }
                }
                if (mem) {
                    uint8_t* ib = static_cast<uint8_t*>(a->image) + image_balance_offset(W, H, NB);
                    bb.hdr = reinterpret_cast<ViewCostHdr*>(mem);
                    bb.cost = reinterpret_cast<uint16_t*>(mem + hdr_bytes);
                    bb.cur = reinterpret_cast<uint32_t*>(ib);
                    bb.perm = reinterpret_cast<uint16_t*>(ib + 256);
                    bb.vm = a->viewmatrix; bb.pt = a->points_transform; bb.view_id = (long long)a->view_id;
                    bb.tol = 1e-6f * (float)g_view_pose_tol_e6;
                    bb.items = items; bb.nslots4 = nslots4; bb.W = W; bb.H = H;
                    // the list cut: the plain direct binning behind a global depth sort, one model, the default blend
                    if (direct && !tsort && db.NS == 1 && db.NC <= 16000 && T == db.T && (g_list_cut == 2 || (g_list_cut == 1 && N >= g_list_cut_min_n))) {
                        uint8_t* dm = fs + align256(L.bytes);
                        lc.key = reinterpret_cast<uint32_t*>(mem + vc_cut);
                        lc.owner_n = reinterpret_cast<uint32_t*>(mem + vc_own);
                        lc.stats = &bb.hdr->pad[2];
                        lc.cur = bb.cur;
                        lc.chunk = reinterpret_cast<uint16_t*>(dm + dbs.cut_chunk);
                        lc.tend = reinterpret_cast<uint32_t*>(dm + dbs.cut_tend);
                        lc.flag = reinterpret_cast<uint32_t*>(dm + dbs.cut_flag);
                        lc.ctl = reinterpret_cast<uint32_t*>(dm + dbs.cut_ctl);
                        lc.bkey = reinterpret_cast<uint32_t*>(dm + dbs.cut_bkey);
                        lc.hist = reinterpret_cast<uint32_t*>(dm + dbs.cut_hist);
                        lc.skey = in_alt ? dkey_alt : dkey;
                        lc.tbase = reinterpret_cast<uint32_t*>(dm + dbs.tbase);
                        lc.margin = 1e-3f * (float)g_list_cut_margin_e3;
                        lc.dmax = std::max(0, g_list_cut_deep);
                        lc.enable = 1;
                    }
                }
            }
        }
        if (direct) {
            uint8_t* dm = fs + align256(L.bytes);
            db.M = reinterpret_cast<uint16_t*>(dm + dbs.M); db.GT = reinterpret_cast<uint32_t*>(dm + dbs.GT); db.tbase = reinterpret_cast<uint32_t*>(dm + dbs.tbase);
            db.bsum = reinterpret_cast<uint32_t*>(dm + dbs.bsum);
            zj.p[0] = nullptr; zj.words[0] = 0u;   // (the scatter writes every tile's range; there is no sort scratch)
            zj.p[2] = nullptr; zj.words[2] = 0u;
        }
        if (early_r) launch_counts(bb, zj, window_overflow_, rider.seq, false);
        else launch_counts(bb, zj, window_overflow_, ++pin.s->seq);
    }
    GSR_HIP(hipGetLastError());

    if (speculative) {
        // The host needs R: it polls the pinned word the scan kernel writes (seen ~1 us after the store).  No event is recorded behind
        // the scan any more (round 4): an event record is a barrier packet with a completion signal, and the queue idled ~6 us at it
        // on EVERY forward (tools/api_timeline.sh) for the sake of a fallback.  The poll is bounded by wall time ("poll_iters" x ~50 ns,
        // default 20 ms: a device that far behind, or one that has faulted, is waited for with hipStreamSynchronize, which reports it).
        // "poll_iters" 0 keeps the old protocol: event record + hipEventSynchronize, no busy waiting.
        const bool use_event = g_poll_iters <= 0;
        if (use_event) GSR_HIP(hipEventRecord(pin.s->ev, st));
        rc = launch_binning(cap, total, true);
        if (rc) return rc;
        rc = launch_blend(true);
        if (rc) return rc;
        {
            const auto w0 = std::chrono::steady_clock::now();
            volatile unsigned long long* hp = pin.s->host;
            const unsigned long long want = pin.s->seq;
            bool seen = false;
            if (!use_event) {
                const auto limit = w0 + std::chrono::nanoseconds((long long)g_poll_iters * 50);
                while (!(seen = (hp[1] == want))) {
                    for (int it = 0; it < 64 && !(seen = (hp[1] == want)); it++) cpu_relax();
                    if (seen || std::chrono::steady_clock::now() > limit) break;
                }
            }
            if (!seen) {
                if (use_event) GSR_HIP(hipEventSynchronize(pin.s->ev));
                else GSR_HIP(hipStreamSynchronize(st));
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            g_fwd_wait_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - w0).count();
        }
        g_spec_forwards++;
    } else {
        const auto w0 = std::chrono::steady_clock::now();
        GSR_HIP(hipStreamSynchronize(st));
        g_fwd_wait_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - w0).count();
        g_exact_forwards++;
    }
    R = *static_cast<volatile unsigned long long*>(pin.s->host);
    {   // a look-back of one of the radix sorts gave up (since the last call that looked): whatever was sorted is garbage
        static std::map<int, unsigned long long> seen_giveups;   // per device; guarded by g_state_mutex
        const unsigned long long gv = static_cast<volatile unsigned long long*>(pin.s->host)[2];
        bool fresh;
        {
            std::lock_guard<std::mutex> lk(g_state_mutex);
            unsigned long long& seen = seen_giveups[dev_id];
            fresh = gv != seen;
            seen = gv;
        }
        if (fresh)
            return fail(GSR_ERR_HIP, "a radix-sort look-back gave up (status words overwritten?): the binning of this or the previous forward is invalid%s");
        if (early_r) {   // R and the give-up counter came from the rider, in front of this forward's own sorts: checked again behind them
            pin.s->pending = true; pin.s->pending_seq = rider.seq; pin.s->pending_R = R + (unsigned long long)g_debug_late_bias; pin.s->pending_giveups = gv;
            g_debug_late_bias = 0;
        }
    }
    bool resorted = false;
    if (wide_depth && static_cast<volatile unsigned long long*>(pin.s->host)[3] != 0ull) {
        // a depth key beyond the 27-bit window of the three-pass sort (a visible Gaussian farther than 13 107 units): its clamped digit
        // ordered it by index among its like.  Sort again on all 32 bits -- from the clamped result, which kept equal keys in index
        // order, so the stable full sort of it IS the sort -- recount, rescan, and take the exact flow; this caller stays on the
        // four-pass sort from now on.
        { std::lock_guard<std::mutex> lk(g_state_mutex); g_full_depth_sort[hint_key] = true; }
        g_depth_window_resorts++;
        uint32_t *k0 = in_alt ? dkey_alt : dkey, *v0 = in_alt ? gid_alt : gid, *k1 = in_alt ? dkey : dkey_alt, *v1 = in_alt ? gid : gid_alt;
        int alt2 = 0;
        GSR_HIP(onesweep_sort_pairs<uint32_t>(k0, v0, k1, v1, (uint32_t)N, 0, 32, depth_scratch, &alt2, st));
        sorted_gid = alt2 ? v1 : v0;
        launch_counts(BlendBalance{}, ZeroJobs{}, nullptr, ++pin.s->seq);
        GSR_HIP(hipStreamSynchronize(st));
        std::atomic_thread_fence(std::memory_order_acquire);
        R = *static_cast<volatile unsigned long long*>(pin.s->host);
        resorted = true;
    }
    if (R > 0xfffffff0ull) return fail(GSR_ERR_RANGE, "more than 2^32 instances%s");
    if (!speculative || R > cap || resorted) {   // exact flow, or the capacity was too small (the truncated result is overwritten)
        if (speculative && !resorted) g_spec_overflows++;
        rc = alloc_binning(R);
        if (rc) return rc;
        rc = launch_binning(R, nullptr, false);
        if (rc) return rc;
        rc = launch_blend(false);
        if (rc) return rc;
    }
    {
        // next capacity of THIS caller (device, image size, N bucket): 1.25x this frame's count, but never much below what
        // its recent frames needed (views alternate in training, so the hint decays slowly instead of following every
        // small frame down)
        std::lock_guard<std::mutex> lk(g_state_mutex);
        uint64_t& h = g_hints[hint_key];
        h = std::max<uint64_t>(std::max<uint64_t>(R + R / 4, h - h / 32), 1u << 16);
        if (g_hints.size() > 4096) g_hints.clear();   // (a runaway number of distinct callers: start over)
    }
    out->num_rendered = (int64_t)R;
    out->binning = bin;
    out->binning_bytes = B.bytes;
    out->binning_capacity = (int64_t)((speculative && R <= cap && !resorted) ? cap : R);
    out->forward_flags = pack_fwd_flags(opt_ppt, opt_map, opt_ckpt);
    return GSR_OK;
}

// workgroups of the backward blend the device holds at once (occupancy x compute units), per kernel variant
static int blend_bwd_resident(int has_da, int ppt = 2)
{
    static int cached[2][2] = {{0, 0}, {0, 0}};
    const int v = ppt == 1 ? 1 : 0;
    if (cached[v][has_da]) return cached[v][has_da];
    int dev = 0, cus = 256, per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    hipError_t e;
    if (v) e = has_da ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_blend_bwd1<true>, 256, 0)
                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_blend_bwd1<false>, 256, 0);
    else e = has_da ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_blend_bwd2<true>, 128, 0)
                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_blend_bwd2<false>, 128, 0);
    if (e != hipSuccess || per_cu < 1) per_cu = v ? 6 : 12;
    const char* env = getenv("GSR_BWD_WG_PER_CU");   // (experiments)
    if (env && atoi(env) > 0) per_cu = atoi(env);
    return cached[v][has_da] = per_cu * cus;
}

int gsr_backward(const GsrBackwardArgs* a, void* stream_)
{
    CallTimer call_timer(g_bwd_calls, g_bwd_ns);
    hipStream_t st = (hipStream_t)stream_;
    if (!a) return fail(GSR_ERR_ARG, "null args%s");
    int rc = check_common(a->N, a->M, a->D, a->W, a->H);
    if (rc) return rc;
    const int N = a->N, W = a->W, H = a->H;
    BatchDev bt;
    rc = batch_dev(a->batch, N, H, bt);
    if (rc) return rc;
    const int NB = bt.B;
    {   // early R: a forward's count is held against its scan's own report as soon as that has arrived (no waiting here)
        int dev_id = 0;
        (void)hipGetDevice(&dev_id);
        unsigned long long got = 0, want = 0;
        bool bad = false;
        {
            std::lock_guard<std::mutex> lk(g_state_mutex);
            auto it = g_pin_slots.find(dev_id);
            if (it != g_pin_slots.end())
                for (PinSlot* s : it->second)
                    if (!s->busy && late_check(s, false, &got, &want)) { bad = true; break; }
        }
        if (bad) {
            char msg[200];
            snprintf(msg, sizeof msg, "a forward's binning was invalid: early instance count %llu, its scan counted %llu (or a sort look-back gave up)%%s", want, got);
            return fail(GSR_ERR_HIP, msg);
        }
    }
    // the forward's kernel variant / tile map / checkpoint layout travel with its output (forward_flags); a caller of the
    // round-1 ABI (flags 0) gets the process-wide options as before
    int f_ppt = g_blend_ppt ? g_blend_ppt : 7, f_map = g_tile_map, f_ckpt = g_ckpt_first;
    if (a->forward_flags & 1) {
        f_ppt = (int)((a->forward_flags >> 1) & 7); f_map = (int)((a->forward_flags >> 4) & 3); f_ckpt = (int)((a->forward_flags >> 6) & 127);
        if (f_ppt < 1 || f_ppt > 7 || f_map > 2 || f_ckpt < 1) return fail(GSR_ERR_ARG, "forward_flags do not come from gsr_forward%s");
    }
    (void)f_ppt;
    if (N == 0) {
        if (a->d_viewmatrix) GSR_HIP(hipMemsetAsync(a->d_viewmatrix, 0, 64 * NB, st));
        if (a->d_projmatrix) GSR_HIP(hipMemsetAsync(a->d_projmatrix, 0, 64 * NB, st));
        if (a->d_campos) GSR_HIP(hipMemsetAsync(a->d_campos, 0, 12 * NB, st));
        if (a->d_points_transform) GSR_HIP(hipMemsetAsync(a->d_points_transform, 0, 48 * NB, st));
        return GSR_OK;
    }
    if (!a->geom || !a->image || !a->binning || !a->scratch || !a->d_means2D)
        return fail(GSR_ERR_ARG, "missing workspace / gradient pointer%s");
    const int tiles_x = (W + kTile - 1) / kTile, tiles_y = (H + kTile - 1) / kTile, T = tiles_x * tiles_y * NB;
    const Splat* splat = static_cast<const Splat*>(a->geom);
    BinLayout B = bin_layout(a->binning_capacity > 0 ? a->binning_capacity : a->num_rendered, W, H, NB);
    const uint8_t* bin = static_cast<const uint8_t*>(a->binning);
    const uint2* ranges = reinterpret_cast<const uint2*>(bin + B.ranges);
    const uint32_t* list = reinterpret_cast<const uint32_t*>(bin + B.list);
    float* gg = static_cast<float*>(a->scratch);
    // (round 3, measured and dropped: the extension allocating this scratch in the FORWARD and clearing it on a side stream behind
    //  an event, so that the 8 us fill runs next to the forward's kernels instead of in front of the backward's -- 0.863-0.873 ms per
    //  step against 0.847-0.855 with the fill here, same box: the concurrent fill takes more from the sorts than it saves)
    // prepare in backward: the digit counters of the next forward's depth sort live in the hand-over buffer; they are cleared by
    // the prologue's workgroup 0 (no launch of their own) and filled by the per-Gaussian kernel behind the blend
    uint32_t* prep_head = (a->next_view && a->prepared_out) ? reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(a->prepared_out) + prep_layout(N).sort) : nullptr;
    bool prep_head_cleared = false;
    // (the fixed-order debug mode runs the two-pixel kernel whatever the option says: its two waves add their partials into the tile's LDS
    //  row commutatively, the four waves of k_blend_bwd1 do not -- a + b + c + d depends on the order the LDS atomics land in)
    const int bwd_ppt = g_deterministic ? 2 : (g_bwd_ppt ? g_bwd_ppt : 2);
    const bool blend_items = a->num_rendered > 0 && (bwd_ppt == 2 || bwd_ppt == 1);
    // ONE launch clears the per-Gaussian accumulators and (workgroup 0) builds the backward blend's work items from the forward's
    // staged depths -- where the 48 N-byte memset stood
    BwdItemHdr* item_hdr = nullptr;
    const uint2* items = nullptr;
    int bwd_grid = 0;
    {
        const uint32_t* staged4 = reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(a->image) + image_staged_offset(W, H, NB));
        const int tpad = 8 * slots_per_xcd(f_map, T, tiles_x);
        const int want_split = g_bwd_split ? g_bwd_split : 16;
        const int split_ok = (want_split > 1 && f_ppt >= 5) ? 1 : 0;
        const size_t n4 = (size_t)N * kGG / 4;
        int fill_blocks = (int)std::min<size_t>(4096, (n4 + 255) / 256);
        if (blend_items) {
            // (the binning buffer is this library's workspace: the item lists live behind the checkpoints, see bin_layout)
            uint8_t* binw = const_cast<uint8_t*>(bin);
            item_hdr = reinterpret_cast<BwdItemHdr*>(binw + B.items);
            items = reinterpret_cast<const uint2*>(binw + B.items + kItemHdrBytes);
            // persistent grid: what the chip holds at once (never more workgroups than there can be items), a multiple of 8
            const int64_t bound = (a->num_rendered >> 7) + T;
            const int resident = blend_bwd_resident((a->grad_depth || a->grad_alpha) ? 1 : 0, bwd_ppt);
            bwd_grid = (int)std::max<int64_t>(8, (std::min<int64_t>(resident, bound) + 7) / 8 * 8);
        }
        const uint32_t list_cap = (uint32_t)bwd_list_cap(a->binning_capacity > 0 ? a->binning_capacity : a->num_rendered, (size_t)T);
        hipLaunchKernelGGL(k_bwd_prologue, dim3(fill_blocks + (blend_items ? 8 : 0)), dim3(256), 0, st, reinterpret_cast<float4*>(gg), n4,
                           staged4, T, tiles_x, f_map, tpad, f_ckpt, split_ok, item_hdr, const_cast<uint2*>(items), list_cap,
                           (uint32_t)(bwd_grid / 8), blend_items ? prep_head : nullptr, (int)kOnesweepHeadWordsMax);
        prep_head_cleared = blend_items && prep_head != nullptr;
    }
    if (a->num_rendered > 0) {
        const int ppt = bwd_ppt;
        if (NB > 1 && ppt != 2 && ppt != 1) return fail(GSR_ERR_ARG, "batch: served by the persistent backward blend kernels only%s");
        const float* img = static_cast<const float*>(a->image);
        ProfScope ps(P_BLEND_BWD, st);
        if (ppt == 2 || ppt == 1) {
            // (checkpoints are written by k_blend_fwd_w only)
            const float* ckpt = reinterpret_cast<const float*>(bin + B.ckpt);
            // deterministic debug mode: R-sized slots + a sort of the instance positions by Gaussian id (stream-ordered
            // allocations of the library's own: gsr_backward has no allocator callback and this is not a hot path)
            float* det_part = nullptr;
            uint8_t* det_mem = nullptr;
            const uint32_t Rn = (uint32_t)a->num_rendered;
            size_t o_part = 0, o_k0 = 0, o_k1 = 0, o_v0 = 0, o_v1 = 0, o_sort = 0, det_bytes = 0;
            if (g_deterministic) {
                size_t o = 0;
                o_part = o; o += align256((size_t)Rn * kDetStride * 4);
                o_k0 = o; o += align256((size_t)Rn * 4); o_k1 = o; o += align256((size_t)Rn * 4);
                o_v0 = o; o += align256((size_t)Rn * 4); o_v1 = o; o += align256((size_t)Rn * 4);
                o_sort = o; o += radix_scratch_bytes(Rn);
                det_bytes = o;
                GSR_HIP(hipMallocAsync((void**)&det_mem, det_bytes, st));
                det_part = reinterpret_cast<float*>(det_mem + o_part);
                GSR_HIP(hipMemsetAsync(det_part, 0, (size_t)Rn * kDetStride * 4, st));
            }
            BlendBwdArgs ba = {W, H, tiles_x, tiles_y, T, f_map, f_ckpt, 0, ranges, list, splat, a->bg, img, a->grad_color, a->grad_depth,
                               a->grad_alpha, gg, ckpt, det_part, item_hdr, items};
            if (ppt == 1) {
                if (a->grad_depth || a->grad_alpha) hipLaunchKernelGGL(k_blend_bwd1<true>, dim3(bwd_grid), dim3(256), 0, st, ba);
                else hipLaunchKernelGGL(k_blend_bwd1<false>, dim3(bwd_grid), dim3(256), 0, st, ba);
            } else if (a->grad_depth || a->grad_alpha) hipLaunchKernelGGL(k_blend_bwd2<true>, dim3(bwd_grid), dim3(128), 0, st, ba);
            else hipLaunchKernelGGL(k_blend_bwd2<false>, dim3(bwd_grid), dim3(128), 0, st, ba);
            if (g_deterministic) {
                uint32_t* k0 = reinterpret_cast<uint32_t*>(det_mem + o_k0); uint32_t* k1 = reinterpret_cast<uint32_t*>(det_mem + o_k1);
                uint32_t* v0 = reinterpret_cast<uint32_t*>(det_mem + o_v0); uint32_t* v1 = reinterpret_cast<uint32_t*>(det_mem + o_v1);
                hipLaunchKernelGGL(k_det_iota, dim3((Rn + 255) / 256), dim3(256), 0, st, Rn, v0, list, k0, (uint32_t)N);
                int nbits = 1;
                while ((1ll << nbits) < (long long)N) nbits++;
                int in_alt = 0;
                GSR_HIP(radix_sort_pairs<uint32_t>(k0, v0, k1, v1, Rn, 0, ((nbits + 7) / 8) * 8, det_mem + o_sort, &in_alt, st));
                hipLaunchKernelGGL(k_det_reduce, dim3((Rn + 255) / 256), dim3(256), 0, st, Rn, in_alt ? k1 : k0, in_alt ? v1 : v0, det_part, gg);
                GSR_HIP(hipFreeAsync(det_mem, st));
            }
        }
        else return fail(GSR_ERR_ARG, "unknown backward blend variant%s");
    }
    CamParams cp = {a->viewmatrix, a->projmatrix, a->campos, a->tanfovx, a->tanfovy, a->scale_modifier, W, H, a->D, a->M, a->points_transform, bt};
    const int grid = (N + kPreThreads - 1) / kPreThreads;
    const bool want_cam = a->d_viewmatrix || a->d_projmatrix || a->d_campos || a->d_points_transform;
    float* cam_partial = reinterpret_cast<float*>(static_cast<uint8_t*>(a->scratch) + align256((size_t)N * kGG * 4));
    AdamDev ad = {};
    const GsrFusedAdam* fa = a->fused_adam;
    if (fa) {
        if (!a->raw_params || !a->shs || !a->shs_rest || a->cov3D_precomp || a->colors_precomp || !a->scales || !a->rotations || !a->opacities ||
            a->M < 1 || fa->step <= 0)
            return fail(GSR_ERR_ARG, "fused_adam needs raw_params with shs (f_dc) + shs_rest, scales, rotations and opacities%s");
        // f_rest without moment buffers: the group is skipped (GsrFusedAdam) -- only where its gradient is identically zero
        const bool skip_rest = !fa->exp_avg[2] && !fa->exp_avg_sq[2];
        if (skip_rest && (a->D != 0 || (a->next_view && a->next_view->D != 0)))
            return fail(GSR_ERR_ARG, "fused_adam: the f_rest group may be skipped only at sh_degree 0 (this render and the prepared one)%s");
        for (int q = 0; q < 6; q++) {
            ad.inv_bc2s[q] = 1.f;
            if (q == 2 && skip_rest) { ad.m[q] = nullptr; ad.v[q] = nullptr; ad.step_size[q] = 0.f; continue; }
            if (!fa->exp_avg[q] || !fa->exp_avg_sq[q]) return fail(GSR_ERR_ARG, "fused_adam: missing moment buffer%s");
            ad.m[q] = fa->exp_avg[q]; ad.v[q] = fa->exp_avg_sq[q];
            // a group may be steps behind the others (the reference drops the opacity group's update on an opacity-reset
            // iteration): its bias corrections use its own count
            const int64_t tq = fa->step - (int64_t)fa->step_lag[q];
            if (fa->step_lag[q] < 0 || tq < 1) return fail(GSR_ERR_ARG, "fused_adam: step_lag out of range%s");
            ad.step_size[q] = fa->lr[q] / (float)(1.0 - pow((double)fa->beta1, (double)tq));
            ad.inv_bc2s[q] = 1.f / (float)sqrt(1.0 - pow((double)fa->beta2, (double)tq));
            // deferred application: the updated rows go to buffers of their own and the caller adopts them (or not) later
            const float* pin[6] = {a->means3D, a->shs, a->shs_rest, a->opacities, a->scales, a->rotations};
            if (fa->param_out[q] || fa->exp_avg_out[q] || fa->exp_avg_sq_out[q]) {
                if (!fa->param_out[q] || !fa->exp_avg_out[q] || !fa->exp_avg_sq_out[q]) return fail(GSR_ERR_ARG, "fused_adam: a group's three output buffers come together%s");
                if (a->next_view) return fail(GSR_ERR_ARG, "fused_adam: deferred application (param_out) cannot prepare a next view%s");
                ad.dp[q] = (long long)((const char*)fa->param_out[q] - (const char*)pin[q]);
                ad.dm[q] = (long long)((const char*)fa->exp_avg_out[q] - (const char*)fa->exp_avg[q]);
                ad.dv[q] = (long long)((const char*)fa->exp_avg_sq_out[q] - (const char*)fa->exp_avg_sq[q]);
            }
        }
        ad.b1 = fa->beta1; ad.b2 = fa->beta2; ad.eps = fa->eps;
    } else if (!a->d_means3D || !a->d_opacities)
        return fail(GSR_ERR_ARG, "missing workspace / gradient pointer%s");
    DensDev ds = {};
    if (a->densify_stats) {
        const GsrDensifyStats* q = a->densify_stats;
        if (!q->radii || !q->xyz_gradient_accum || !q->denom || !q->max_radii2D) return fail(GSR_ERR_ARG, "densify_stats: four pointers expected%s");
        ds.radii = q->radii; ds.grad_accum = q->xyz_gradient_accum; ds.denom = q->denom; ds.max_radii = q->max_radii2D;
    }
    // "prepare in backward": the next render's preprocess rides in the per-Gaussian kernel (see GsrNextView)
    PrepOut po = {};
    const GsrNextView* nv = a->next_view;
    if (nv) {
        if (!fa || !gsr_prepare_supported(a->M, a->D, a->raw_params) || nv->D < a->D || nv->D > a->D + 1 || nv->D > 3 || !a->prepared_out ||
            !nv->viewmatrix || !nv->projmatrix || !nv->campos || nv->W <= 0 || nv->H <= 0 ||
            ((((uintptr_t)a->shs | (uintptr_t)a->shs_rest) & 15) != 0))
            return fail(GSR_ERR_ARG, "next_view needs fused_adam, raw_params, M = 16, a next degree equal to this render's or one above, "
                                     "16-byte aligned SH tensors and a prepared_out buffer%s");
        const PrepLayout PL = prep_layout(N);
        uint8_t* pb = static_cast<uint8_t*>(a->prepared_out);
        BatchDev nbt = bt;   // the next render is of the same batch (its cameras are arrays of NB entries as well)
        if (NB > 1 && (long long)NB * ((nv->H + kTile - 1) / kTile) > 4095) return fail(GSR_ERR_RANGE, "batch: more than 4095 tile rows in the next view%s");
        po.cp = {nv->viewmatrix, nv->projmatrix, nv->campos, nv->tanfovx, nv->tanfovy, nv->scale_modifier, nv->W, nv->H, nv->D, a->M, nv->points_transform, nbt};
        po.splat = reinterpret_cast<Splat*>(pb + PL.splat);
        po.radii = reinterpret_cast<int32_t*>(pb + PL.radii);
        po.dkey = reinterpret_cast<uint32_t*>(pb + PL.dkey);
        po.gid = reinterpret_cast<uint32_t*>(pb + PL.gid);
        po.rec = reinterpret_cast<TileRec*>(pb + PL.rec);
        po.early_parts = reinterpret_cast<uint2*>(pb + PL.early);
        po.early_window = N > g_prep_hist_max_n ? (1u << 27) - 1u : 0xffffffffu;   // (counted whenever the next forward MAY sort on the 27-bit window)
        if (!prep_head_cleared) GSR_HIP(hipMemsetAsync(prep_head, 0, kOnesweepHeadWordsMax * sizeof(uint32_t), st));   // (no blend launch: empty frame / other variant)
        if (prep_counts_digits(N)) {
            po.dh.ghist = prep_head;
            po.dh.status = onesweep_status(pb + PL.sort);
            po.dh.status_words = onesweep_status_words<uint32_t>((uint32_t)N, 32);
            const int tail_blocks = ((N - (grid - 1) * kPreThreads) % 4 == 0) ? grid : grid - 1;   // (= last_lin below)
            po.dh.clear_threads = (uint32_t)tail_blocks * kPreThreads;
        }
    }
#define GSR_PREB_(DEG, RAW, CAM, ADAM, PREP)                                                                                                \
    hipLaunchKernelGGL((k_preprocess_bwd<DEG, RAW, CAM, ADAM, PREP>), dim3(grid), dim3(kPreThreads), 0, st, cp, N, a->means3D, a->scales,  \
                       a->rotations, a->cov3D_precomp, a->shs, a->shs_rest, a->opacities, ad, splat, gg, a->d_means3D, a->d_means2D,        \
                       a->d_opacities, a->d_colors_precomp, a->d_shs, a->d_shs_rest, a->d_scales, a->d_rotations, a->d_cov3D_precomp,      \
                       cam_partial, po, ds)
#define GSR_PREB(DEG)                                                     \
    do {                                                                  \
        if (fa) { if (want_cam) GSR_PREB_(DEG, true, true, true, -1); else GSR_PREB_(DEG, true, false, true, -1); }                 \
        else if (a->raw_params) { if (want_cam) GSR_PREB_(DEG, true, true, false, -1); else GSR_PREB_(DEG, true, false, false, -1); }   \
        else { if (want_cam) GSR_PREB_(DEG, false, true, false, -1); else GSR_PREB_(DEG, false, false, false, -1); }               \
    } while (0)
#define GSR_PREB_NEXT(DEG, NDEG) do { if (want_cam) GSR_PREB_(DEG, true, true, true, NDEG); else GSR_PREB_(DEG, true, false, true, NDEG); } while (0)
#define GSR_PRE_TAIL(NDEG)                                                                                                               \
    hipLaunchKernelGGL((k_preprocess<NDEG, true>), dim3(1), dim3(kPreThreads), 0, st, po.cp, N, a->means3D, a->scales, a->rotations,       \
                       (const float*)nullptr, a->opacities, a->shs, a->shs_rest, (const float*)nullptr, po.splat, po.radii, po.dkey,     \
                       po.gid, po.rec, (uint32_t*)nullptr, 0, grid - 1, po.dh, po.early_parts, po.early_window)
    {
        ProfScope ps(P_PRE_BWD, st);
        if (nv) {   // (this render's degree, the next render's): equal, or one step up (checked above)
            switch (a->D * 4 + nv->D) {
                case 0: GSR_PREB_NEXT(0, 0); break;
                case 1: GSR_PREB_NEXT(0, 1); break;
                case 5: GSR_PREB_NEXT(1, 1); break;
                case 6: GSR_PREB_NEXT(1, 2); break;
                case 10: GSR_PREB_NEXT(2, 2); break;
                case 11: GSR_PREB_NEXT(2, 3); break;
                default: GSR_PREB_NEXT(3, 3); break;
            }
            // a ragged last block whose rows do not form whole 16-byte vectors takes the kernel's general (non-linear) tile
            // path, which has no next-view tail: the ordinary preprocess runs on that one block, on the updated parameters
            const int nlast = N - (grid - 1) * kPreThreads;
            const bool last_lin = (nlast % 4) == 0;   // (the tensors' 16-byte alignment was checked above)
            if (!last_lin) {
                switch (nv->D) {
                    case 0: GSR_PRE_TAIL(0); break;
                    case 1: GSR_PRE_TAIL(1); break;
                    case 2: GSR_PRE_TAIL(2); break;
                    default: GSR_PRE_TAIL(3); break;
                }
            }
        } else {
        switch (a->shs ? a->D : 0) {
            case 0: GSR_PREB(0); break;
            case 1: GSR_PREB(1); break;
            case 2: GSR_PREB(2); break;
            default: GSR_PREB(3); break;
        }
        }
    }
#undef GSR_PRE_TAIL
#undef GSR_PREB_NEXT
#undef GSR_PREB
#undef GSR_PREB_
    if (want_cam)
        hipLaunchKernelGGL(k_cam_reduce, dim3(kCamVals, NB), dim3(256), 0, st, cam_partial, grid, a->d_viewmatrix, a->d_projmatrix, a->d_campos,
                           a->d_points_transform, bt);
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}

int gsr_debug_view_cache_stats(int32_t W, int32_t H, int64_t out[4])
{
    // the balanced placement's per-view cost caches of the CURRENT device for this frame size: lookups, hits, entries in use, caches
    int dev_id = 0;
    if (hipGetDevice(&dev_id) != hipSuccess) return fail(GSR_ERR_HIP, "hipGetDevice failed%s");
    out[0] = out[1] = out[2] = out[3] = 0;
    std::vector<uint8_t*> mems;
    {
        std::lock_guard<std::mutex> lk(g_state_mutex);
        for (auto& c : g_view_costs) if (c.dev == dev_id && c.W == W && c.H == H) mems.push_back(c.mem);
    }
    for (uint8_t* m : mems) {
        ViewCostHdr h;
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&h, m, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess)
            return fail(GSR_ERR_HIP, "view cache read-back failed%s");
        out[0] += h.pad[0]; out[1] += h.pad[1]; out[3] += 1;
        for (int i = 0; i < kVcEntries; i++) out[2] += h.key[i] != 0ull;
    }
    return GSR_OK;
}

int gsr_debug_list_cut_stats(int32_t W, int32_t H, int64_t out[5])
{
    // the list cut's counters of the CURRENT device for this frame size (ListCut::stats): renders that ran with the cut machinery,
    // renders whose repair pass found flagged tiles, tiles repaired, chunks of the depth order behind the frame's cut (summed), deep
    // tiles (summed)
    int dev_id = 0;
    if (hipGetDevice(&dev_id) != hipSuccess) return fail(GSR_ERR_HIP, "hipGetDevice failed%s");
    out[0] = out[1] = out[2] = out[3] = out[4] = 0;
    std::vector<uint8_t*> mems;
    {
        std::lock_guard<std::mutex> lk(g_state_mutex);
        for (auto& c : g_view_costs) if (c.dev == dev_id && c.W == W && c.H == H) mems.push_back(c.mem);
    }
    for (uint8_t* m : mems) {
        ViewCostHdr h;
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&h, m, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess)
            return fail(GSR_ERR_HIP, "view cache read-back failed%s");
        for (int q = 0; q < 5; q++) out[q] += h.pad[2 + q];
    }
    return GSR_OK;
}

int gsr_debug_direct_binning_geometry(int32_t N, int32_t T, int64_t out[7])
{
    if (!out) return GSR_ERR_ARG;
    DirectBin db = {};
    DirectBinScratch ds = {};
    const bool ok = direct_bin_geometry(N, T, db, ds);
    out[0] = ok ? 1 : 0; out[1] = db.S; out[2] = db.NC; out[3] = db.G; out[4] = db.Cg; out[5] = db.Tp; out[6] = ok ? (int64_t)ds.bytes : 0;
    return GSR_OK;
}

int gsr_debug_read_binning(const void* binning, int64_t binning_capacity, int64_t num_rendered, int32_t W, int32_t H,
                            uint32_t* ranges_out, uint32_t* list_out, void* stream_)
{
    if (!binning || W <= 0 || H <= 0 || num_rendered < 0) return fail(GSR_ERR_ARG, "bad debug_read_binning args%s");
    const BinLayout B = bin_layout(binning_capacity > 0 ? binning_capacity : num_rendered, W, H);
    const size_t T = (size_t)((W + kTile - 1) / kTile) * ((H + kTile - 1) / kTile);
    const uint8_t* bin = static_cast<const uint8_t*>(binning);
    hipStream_t st = (hipStream_t)stream_;
    if (ranges_out) GSR_HIP(hipMemcpyAsync(ranges_out, bin + B.ranges, T * sizeof(uint2), hipMemcpyDeviceToDevice, st));
    if (list_out && num_rendered > 0)
        GSR_HIP(hipMemcpyAsync(list_out, bin + B.list, (size_t)num_rendered * 4, hipMemcpyDeviceToDevice, st));
    return GSR_OK;
}

int64_t gsr_get_counter(const char* name)
{
    if (!name) return -1;
    if (!strcmp(name, "spec_overflows")) return g_spec_overflows.load();
    if (!strcmp(name, "late_checks")) return g_late_checks.load();
    if (!strcmp(name, "late_mismatches")) return g_late_mismatches.load();
    if (!strcmp(name, "depth_window_resorts")) return g_depth_window_resorts.load();
    if (!strcmp(name, "blend_bwd_resident")) return blend_bwd_resident(0);
    if (!strcmp(name, "spec_forwards")) return g_spec_forwards.load();
    if (!strcmp(name, "exact_forwards")) return g_exact_forwards.load();
    if (!strcmp(name, "forward_calls")) return g_fwd_calls.load();
    if (!strcmp(name, "forward_ns")) return g_fwd_ns.load();
    if (!strcmp(name, "forward_wait_ns")) return g_fwd_wait_ns.load();
    if (!strcmp(name, "backward_calls")) return g_bwd_calls.load();
    if (!strcmp(name, "backward_ns")) return g_bwd_ns.load();
    if (!strcmp(name, "spec_callers")) { std::lock_guard<std::mutex> lk(g_state_mutex); return (int64_t)g_hints.size(); }
    return -1;
}

int gsr_mark_visible(int32_t N, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present, void* stream_)
{
    (void)projmatrix;
    if (N < 0 || (N > 0 && (!means3D || !viewmatrix || !present))) return fail(GSR_ERR_ARG, "bad mark_visible args%s");
    if (N == 0) return GSR_OK;
    hipLaunchKernelGGL(k_mark_visible, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream_, N, means3D, viewmatrix, present);
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}

int gsr_sort_pairs_u32(uint32_t* keys, uint32_t* vals, uint32_t* keys_alt, uint32_t* vals_alt, uint32_t n, int begin_bit, int end_bit,
                       void* scratch, size_t scratch_bytes, int* result_in_alt, void* stream)
{
    if (scratch_bytes < radix_scratch_bytes(n) || !result_in_alt) return fail(GSR_ERR_ARG, "sort scratch too small%s");
    if (g_sort_algo && end_bit - begin_bit <= 32)
        GSR_HIP(onesweep_sort_pairs<uint32_t>(keys, vals, keys_alt, vals_alt, n, begin_bit, end_bit, scratch, result_in_alt, (hipStream_t)stream));
    else
        GSR_HIP(radix_sort_pairs<uint32_t>(keys, vals, keys_alt, vals_alt, n, begin_bit, end_bit, scratch, result_in_alt, (hipStream_t)stream));
    return GSR_OK;
}

int gsr_sort_pairs_u16(uint16_t* keys, uint32_t* vals, uint16_t* keys_alt, uint32_t* vals_alt, uint32_t n, int begin_bit, int end_bit,
                       void* scratch, size_t scratch_bytes, int* result_in_alt, void* stream)
{
    if (scratch_bytes < radix_scratch_bytes(n) || !result_in_alt) return fail(GSR_ERR_ARG, "sort scratch too small%s");
    if (g_sort_algo)
        GSR_HIP(onesweep_sort_pairs<uint16_t>(keys, vals, keys_alt, vals_alt, n, begin_bit, end_bit, scratch, result_in_alt, (hipStream_t)stream));
    else
        GSR_HIP(radix_sort_pairs<uint16_t>(keys, vals, keys_alt, vals_alt, n, begin_bit, end_bit, scratch, result_in_alt, (hipStream_t)stream));
    return GSR_OK;
}

}  // extern "C"
