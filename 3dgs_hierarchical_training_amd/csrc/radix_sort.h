// radix_sort.h -- stable LSD radix sort of (key, u32 value) pairs for gfx950, hand-written.
//
// Used twice per forward (DESIGN.md "binning"): 32-bit depth keys over the N Gaussians, then 16-bit tile
// keys over the R instances.  One pass = 8 bits = three launches:
//   k_radix_hist     per-block digit histogram (block = 4096 keys)        -> hist[digit][block]
//   k_radix_scan     one workgroup per digit: exclusive scan over blocks + digit base
//   k_radix_scatter  stable in-block ranking (wave_rank below), LDS-staged so that the global writes of one
//                    digit run are consecutive lanes -> coalesced
// Stability inside a block comes from an explicit (wave, round, lane) order: wave w owns keys
// [w*1024, (w+1)*1024) of the tile, ranks them round by round against its own running per-digit count, and the
// per-wave counts are prefix-summed in wave order.
#pragma once
#include <map>
#include <mutex>
#include <utility>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gsr {

constexpr int kSortThreads = 256;
constexpr int kSortIPT = 16;                                // items per thread
constexpr int kSortTile = kSortThreads * kSortIPT;          // 4096 keys per block
constexpr int kSortWaves = kSortThreads / 64;

__device__ __forceinline__ unsigned long long lanemask_lt()
{
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    return (1ull << lane) - 1ull;
}

// ---- in-wave ranking -------------------------------------------------------------------------------------------
// "Which lanes of my wave hold my digit" without the 8 ballots + 8 per-lane selects of the classic match-any loop
// (that loop was ~60 VALU instructions per key and made the scatter kernels issue-bound at 2 waves per SIMD): every
// lane XORs its lane bit into a per-wave 64-bit LDS word of its digit and reads the word before and after; the
// difference is exactly the set of lanes that toggled it this round.  XOR commutes, so the outcome does not depend
// on the order in which the LDS unit serialises same-address atomics; LDS instructions of one wave execute in issue
// order, so "read, xor, read" needs no barrier; nothing is ever reset (a round only looks at the difference).
// s_cnt[digit] is the wave's running count of the digit over the rounds done so far.
// rnk[r] = number of this wave's keys with the same digit that precede key r in (round, lane) order.
template <int IPT, int NB = 256>
__device__ __forceinline__ void wave_rank(unsigned long long* s_mask, uint32_t* s_cnt, const uint32_t (&dig)[IPT],
                                          uint32_t (&rnk)[IPT], int lane)
{
#pragma unroll
    for (int k = 0; k < NB / 64; k++) { s_mask[k * 64 + lane] = 0ull; s_cnt[k * 64 + lane] = 0u; }
    __builtin_amdgcn_wave_barrier();
    const unsigned long long bit = 1ull << lane, lt = bit - 1ull;
    volatile unsigned long long* vmask = s_mask;
    volatile uint32_t* vcnt = s_cnt;
#pragma unroll
    for (int r = 0; r < IPT; r++) {
        const uint32_t d = dig[r];
        uint32_t below, group;
        const uint32_t base = vcnt[d];
        if (__all(d == (uint32_t)__builtin_amdgcn_readfirstlane((int)d))) {   // one digit in the whole wave (high passes)
            below = (uint32_t)lane; group = 64u;
        } else {
            const unsigned long long before = vmask[d];
            atomicXor(&s_mask[d], bit);
            const unsigned long long peers = before ^ vmask[d];
            below = (uint32_t)__popcll(peers & lt);
            group = (uint32_t)__popcll(peers);
        }
        rnk[r] = base + below;
        __builtin_amdgcn_wave_barrier();
        if (below == 0u) vcnt[d] = base + group;
        __builtin_amdgcn_wave_barrier();
    }
}

// exclusive scan of one value per thread over the block's WAVES * 64 threads (two barriers)
template <int WAVES>
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, uint32_t* s_wsum /*[WAVES]*/, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) s_wsum[wave] = inc;
    __syncthreads();
    uint32_t add = 0;
#pragma unroll
    for (int w = 0; w < WAVES; w++) add += (w < wave) ? s_wsum[w] : 0u;
    __syncthreads();
    return add + inc - v;
}

template <typename KeyT>
__global__ __launch_bounds__(kSortThreads) void k_radix_hist(const KeyT* __restrict__ keys, uint32_t n, int shift,
                                                             uint32_t* __restrict__ hist, uint32_t nblocks,
                                                             uint32_t* __restrict__ totals)
{
    __shared__ uint32_t h[256];
    const int tid = threadIdx.x;
    h[tid] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (uint32_t)kSortTile;
    if (base + kSortTile <= n && (((uintptr_t)(keys + base)) & 15) == 0) {
        // full tile: 16-byte loads (order is irrelevant for a histogram)
        constexpr int KPV = 16 / (int)sizeof(KeyT);                 // keys per 16-byte vector
        constexpr int NV = kSortTile / KPV / kSortThreads;           // vectors per thread
        const uint4* v4 = reinterpret_cast<const uint4*>(keys + base);
#pragma unroll
        for (int r = 0; r < NV; r++) {
            const uint4 q = v4[r * kSortThreads + tid];
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (sizeof(KeyT) == 4) atomicAdd(&h[(w[c] >> shift) & 0xffu], 1u);
                else {
                    atomicAdd(&h[((w[c] & 0xffffu) >> shift) & 0xffu], 1u);
                    atomicAdd(&h[((w[c] >> 16) >> shift) & 0xffu], 1u);
                }
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < kSortIPT; r++) {
            const uint32_t idx = base + r * kSortThreads + tid;
            if (idx < n) atomicAdd(&h[((uint32_t)keys[idx] >> shift) & 0xffu], 1u);
        }
    }
    __syncthreads();
    hist[(uint32_t)tid * nblocks + blockIdx.x] = h[tid];
    if (h[tid]) atomicAdd(&totals[tid], h[tid]);
}

// grid = 256 (one workgroup per digit).  hist[d][*] -> exclusive prefix over blocks, plus the
// exclusive prefix over the totals of digits < d (totals[] accumulated by k_radix_hist).
static __global__ __launch_bounds__(256) void k_radix_scan(uint32_t* __restrict__ hist, uint32_t nblocks,
                                                    const uint32_t* __restrict__ totals)
{
    __shared__ uint32_t s_part[256];
    const int tid = threadIdx.x;
    const int d = blockIdx.x;
    uint32_t digit_base = 0;
    for (int k = 0; k < d; k++) digit_base += totals[k];   // uniform -> scalar loads
    // scan this digit's row: each thread owns a contiguous chunk
    uint32_t* row = hist + (size_t)d * nblocks;
    const uint32_t chunk = (nblocks + 255u) / 256u;
    const uint32_t lo = min(nblocks, (uint32_t)tid * chunk), hi = min(nblocks, lo + chunk);
    uint32_t sum = 0;
    for (uint32_t b = lo; b < hi; b++) sum += row[b];
    s_part[tid] = sum;
    __syncthreads();
    // exclusive scan of the 256 partials (Hillis-Steele in LDS)
    for (int off = 1; off < 256; off <<= 1) {
        uint32_t v = (tid >= off) ? s_part[tid - off] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    uint32_t run = digit_base + s_part[tid] - sum;
    for (uint32_t b = lo; b < hi; b++) {
        const uint32_t c = row[b];
        row[b] = run;
        run += c;
    }
}

template <typename KeyT>
__global__ __launch_bounds__(kSortThreads) void k_radix_scatter(const KeyT* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                                KeyT* __restrict__ kout, uint32_t* __restrict__ vout, uint32_t n,
                                                                int shift, const uint32_t* __restrict__ hist, uint32_t nblocks)
{
    __shared__ unsigned long long s_mask[kSortWaves][256];
    __shared__ uint32_t s_cnt[kSortWaves][256];
    __shared__ KeyT s_keys[kSortTile];
    __shared__ uint32_t s_vals[kSortTile];
    __shared__ uint32_t s_start[256];    // block-local exclusive start of each digit
    __shared__ uint32_t s_gbase[256];    // global base of each digit for this block minus s_start
    __shared__ uint32_t s_wsum[kSortWaves];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t base = blockIdx.x * (uint32_t)kSortTile;
    const uint32_t valid = min((uint32_t)kSortTile, n - base);

    KeyT key[kSortIPT];
    uint32_t val[kSortIPT];
    uint32_t dig[kSortIPT];
    uint32_t rnk[kSortIPT];
#pragma unroll
    for (int r = 0; r < kSortIPT; r++) {
        const uint32_t p = (uint32_t)(wave * (kSortIPT * 64) + r * 64 + lane);   // wave-major: see wave_rank
        if (p < valid) {
            key[r] = kin[base + p];
            val[r] = vin[base + p];
            dig[r] = ((uint32_t)key[r] >> shift) & 0xffu;
        } else {
            key[r] = (KeyT)~(KeyT)0; val[r] = 0u; dig[r] = 255u;   // padding sorts behind every real item
        }
    }
    wave_rank<kSortIPT>(s_mask[wave], s_cnt[wave], dig, rnk, lane);
    __syncthreads();
    uint32_t mine = 0;   // thread = digit: exclusive prefix of the per-wave counts in wave order
#pragma unroll
    for (int w = 0; w < kSortWaves; w++) {
        const uint32_t c = s_cnt[w][tid];
        s_cnt[w][tid] = mine;
        mine += c;
    }
    const uint32_t excl = block_scan_excl<kSortWaves>(mine, s_wsum, tid);
    s_start[tid] = excl;
    s_gbase[tid] = hist[(size_t)tid * nblocks + blockIdx.x] - excl;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSortIPT; r++) {
        const uint32_t lp = s_start[dig[r]] + s_cnt[wave][dig[r]] + rnk[r];
        s_keys[lp] = key[r];
        s_vals[lp] = val[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSortIPT; r++) {
        const uint32_t p = r * kSortThreads + tid;
        if (p < valid) {
            const KeyT k = s_keys[p];
            const uint32_t d = ((uint32_t)k >> shift) & 0xffu;
            const uint32_t g = s_gbase[d] + p;
            kout[g] = k;
            vout[g] = s_vals[p];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Onesweep variant: one launch per 8-bit pass (+ one launch that histograms ALL passes up front).
// Each block takes a ticket (= which 4096-key tile it owns, so lower tiles are always already running), ranks
// its keys exactly as k_radix_scatter does, publishes its per-digit counts and finds its global offsets by
// decoupled look-back over the earlier tiles' status words.  A status word is ONE 32-bit value
// {flag:2, count:30} written and polled with relaxed agent-scope atomics (write-through / L1-bypassing on
// gfx950): the data IS the flag, so no fence is needed (MI355X guide G16, form R2).  Every started block
// publishes its LOCAL count before it waits on anything, so the look-back can always walk back to tile 0:
// no circular wait.  Versus hist+scan+scatter this reads the keys once per pass and saves two launches.
// ------------------------------------------------------------------------------------------------
// A pass runs WITHOUT tickets (blockIdx order = look-back order) when its whole grid is co-resident on the device; the
// bound -- workgroups of this very kernel the device holds at once -- is asked of the runtime per device and kernel
// (occupancy API x the device's CU count: a partitioned (CPX / NPS) device reports fewer CUs and gets a smaller bound),
// never assumed.  Above it, workgroups take tickets, which is correct at any residency.
template <typename KeyT> struct OsCfg { static constexpr int kThreads = 512, kTile = 5120; };
template <> struct OsCfg<uint32_t> { static constexpr int kThreads = 1024, kTile = 4096; };

template <typename KernelT>
inline uint32_t onesweep_resident_blocks(KernelT kernel, int threads)
{
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, uint32_t> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    const auto key = std::make_pair(dev, reinterpret_cast<const void*>(kernel));
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int per_cu = 0, cus = 0;
    uint32_t v = 0;   // unknown: always take tickets
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && per_cu > 0 && cus > 0)
        v = (uint32_t)per_cu * (uint32_t)cus;
    cache[key] = v;
    return v;
}
// Look-backs that gave up (a predecessor never published: the status words were overwritten by a caller's bug).  The sort's result is
// then garbage; the counter lets the host say so: gsr_forward reads it back with the instance count and fails the call.
__device__ unsigned int g_onesweep_giveups = 0u;
struct OsRider;
constexpr int kOsRanges = 32;
constexpr uint32_t kOsLocal = 1u << 30, kOsIncl = 2u << 30, kOsMask = (1u << 30) - 1u;
constexpr uint32_t kOsTicketWords = (4 * kOsRanges + 63) / 64 * 64;   // tickets[pass][run], padded
constexpr uint32_t kOnesweepHeadWords = 4 * kOsRanges * 256 + kOsTicketWords;   // scratch head: ghist[4][256] + tickets (padded); status follows
// Round 4, the depth sort over more than a few hundred thousand keys: THREE passes of 9-bit digits over the 27 bits of
// (key - bias) -- positive floats between the near plane (0.2) and 0.2 x 2^16 = 13 107 -- instead of four of 8 over 32 (76 -> 61 us
// at 1 M keys).  512-entry digit tables: k_onesweep<KeyT, 512> holds 64 + 32 kB of XOR-mask / count tables per 16-wave workgroup,
// one workgroup per CU (the 245 tiles of 1 M keys are co-resident all the same); table stride in memory 512.  A key beyond the
// window is clamped (so the passes stay well defined) and COUNTED in a word of the sort's own (zeroed) scratch head
// (onesweep_overflow_word): the caller then sorts again with the four 8-bit passes -- from the clamped result: a stable sort of
// the full keys of any permutation that kept equal keys in index order is the sort.
constexpr uint32_t kOnesweepHeadWords9 = 3 * kOsRanges * 512 + kOsTicketWords;
constexpr uint32_t kOnesweepHeadWordsMax = kOnesweepHeadWords9 > kOnesweepHeadWords ? kOnesweepHeadWords9 : kOnesweepHeadWords;
// (the 9-bit layout uses 3 x 32 of its 128 ticket words; word 100 of them counts the keys beyond the window)
inline unsigned int* onesweep_overflow_word(void* scratch) { return static_cast<unsigned int*>(scratch) + 3 * kOsRanges * 512 + 100; }

// Geometry of the per-run digit counts (what k_radix_ghist produces), for kernels that already hold the keys in registers
// and count the digits themselves -- the producer of the keys then replaces the histogram launch (hist_done below):
//   run of key position o  = o / onesweep_run_len<KeyT>(n)          (n = number of keys, as the sort will see it)
//   digit p of key k       = ((k >> begin_bit) >> (dbits p)) & ((1 << min(dbits, bits - dbits p)) - 1), dbits = onesweep_dbits(bits)
//   counter                = ghist[(p * kOsRanges + run) * 256 + digit]    (ghist = head of the scratch, zero before the first add)
// and it clears the status words behind the head: onesweep_status_words<KeyT>(capacity, bits) 32-bit words at onesweep_status(scratch).
template <typename KeyT>
__host__ __device__ inline uint32_t onesweep_run_len(uint32_t n)
{
    constexpr uint64_t kT = (uint64_t)OsCfg<KeyT>::kTile;
    const uint64_t ntiles = ((uint64_t)n + kT - 1) / kT, per = (ntiles + kOsRanges - 1) / kOsRanges;
    return (uint32_t)(per * kT);   // (<= n / 32 + 2 kT: no overflow)
}
__host__ __device__ inline int onesweep_passes(int bits) { return (bits + 7) / 8; }
__host__ __device__ inline int onesweep_dbits(int bits) { const int p = onesweep_passes(bits); return (bits + p - 1) / p; }
template <typename KeyT>
__host__ __device__ inline uint32_t onesweep_status_words(uint32_t capacity, int bits)
{
    const uint32_t nblocks = (capacity + OsCfg<KeyT>::kTile - 1) / OsCfg<KeyT>::kTile;
    return (uint32_t)((size_t)onesweep_passes(bits) * nblocks * 256);
}
__host__ __device__ inline uint32_t* onesweep_status(void* scratch) { return static_cast<uint32_t*>(scratch) + kOnesweepHeadWords; }

// A rider on the histogram kernel (round 5): the kernel that PRODUCED the keys left per-block partial sums of something the host is
// waiting for (gsr_forward: the instance count R = the sum of the Gaussians' tile counts, which does not depend on the order this sort
// is about to establish); the histogram kernel's last block adds them up and publishes the total to pinned host memory -- a kernel
// boundary after the producer, no launch of its own, and the host has its number ~10 us into the forward instead of behind the sort,
// the tile counts and the scan.  host: [0] sum of parts.x, [1] seq (written last), [2] look-backs that gave up so far, [3] sum of parts.y.
struct OsRider {
    const uint2* parts;
    uint32_t nparts;
    unsigned long long* host;   // nullptr: no rider
    unsigned long long seq;
};
__device__ __forceinline__ void onesweep_rider_publish(const OsRider& rd, unsigned long long (*s_r)[16]);

#ifdef GSR_OS_TIMING
__device__ unsigned long long g_gh_dbg[512 * 8];   // [workgroup of the last histogram launch][phase]
#endif
// GS = stride of the digit tables (256; 512 for 9-bit digits).  bias / kclamp: the digit is taken of min(key - bias, kclamp)
// (0 / all ones: of the key); a key beyond the window that is not the all-ones padding key is counted in *overflow.
template <typename KeyT, int PASSES, int GS = 256>
__global__ __launch_bounds__(256) void k_radix_ghist(const KeyT* __restrict__ keys, uint32_t n, int begin_bit,
                                                     uint32_t* __restrict__ ghist /*[PASSES][runs][GS]*/,
                                                     const unsigned long long* __restrict__ n_dev,
                                                     uint32_t* __restrict__ status, uint32_t status_words, int dbits, int bits, uint32_t bias = 0u,
                                                     uint32_t kclamp = 0xffffffffu, unsigned int* __restrict__ overflow = nullptr, OsRider rider = OsRider{})
{
#ifdef GSR_OS_TIMING
    unsigned long long ght[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ght_last = __builtin_readcyclecounter();
    ght[6] = ght_last;
#define GH_T(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); ght[k] += now_ - ght_last; ght_last = now_; } while (0)
#else
#define GH_T(k) do { } while (0)
#endif
    if (rider.host && blockIdx.x == gridDim.x - 1) {   // (block-uniform)
        __shared__ unsigned long long s_r[2][16];
        onesweep_rider_publish(rider, s_r);
    }
    GH_T(0);   // rider (last block only)
    // digit p covers key bits [dbits p, min(dbits (p + 1), bits)) above begin_bit (<= 8 wide: the tables keep 256 entries)
    uint32_t dmask[PASSES];
#pragma unroll
    for (int p = 0; p < PASSES; p++) dmask[p] = (1u << min(dbits, bits - dbits * p)) - 1u;
    if (n_dev) n = (uint32_t)min((unsigned long long)n, *n_dev);   // device-side count (grid sized for a capacity)
    // clear the look-back status words of all passes (they are first touched by the pass kernels that follow)
    for (uint32_t q = blockIdx.x * 256 + threadIdx.x; q < status_words / 4; q += gridDim.x * 256)
        reinterpret_cast<uint4*>(status)[q] = make_uint4(0u, 0u, 0u, 0u);
    __shared__ uint32_t h[PASSES][GS];
    const int tid = threadIdx.x;
    bool beyond = false;
#pragma unroll
    for (int p = 0; p < PASSES; p++)
        for (int d = tid; d < GS; d += 256) h[p][d] = 0;
    __syncthreads();
    GH_T(1);   // status clear issued, tables zeroed
    constexpr uint64_t kT = (uint64_t)OsCfg<KeyT>::kTile;
    const uint64_t ntiles = ((uint64_t)n + kT - 1) / kT, per = (ntiles + kOsRanges - 1) / kOsRanges;
    const uint32_t bpr = gridDim.x / kOsRanges, x = blockIdx.x / bpr, sub = blockIdx.x - x * bpr;
    const uint64_t lo = min((uint64_t)n, (uint64_t)x * per * kT), hi = min((uint64_t)n, (uint64_t)(x + 1) * per * kT);
    constexpr int KPV = 16 / (int)sizeof(KeyT);   // keys per 16-byte load
    const KeyT* kr = keys + lo;
    const uint32_t len = (uint32_t)(hi - lo);
    const uint32_t nv = (((uintptr_t)kr & 15) == 0) ? len / KPV : 0u;
    const uint4* k4 = reinterpret_cast<const uint4*>(kr);
    // four 16-byte loads in flight per thread (round 5: the loop's trip count is a run-time value, hipcc does not unroll it, and each
    // turn was a memory round trip of its own in a kernel that has four of them per thread at 1 M keys)
    for (uint32_t i0 = sub * 256 + tid; i0 < nv; i0 += 4u * bpr * 256u) {
        uint4 qq[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = i0 + (uint32_t)u * bpr * 256u;
            qq[u] = k4[i < nv ? i : i0];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (i0 + (uint32_t)u * bpr * 256u >= nv) break;
            const uint32_t w[4] = {qq[u].x, qq[u].y, qq[u].z, qq[u].w};
#pragma unroll
            for (int c = 0; c < 4; c++) {
#pragma unroll
                for (int e = 0; e < KPV / 4; e++) {
                    const uint32_t kraw = (uint32_t)(sizeof(KeyT) == 4 ? w[c] : ((w[c] >> (16 * e)) & 0xffffu));
                    beyond |= (kraw - bias > kclamp) && kraw != (uint32_t)(KeyT)~(KeyT)0;
                    const uint32_t k = min(kraw - bias, kclamp) >> begin_bit;
#pragma unroll
                    for (int p = 0; p < PASSES; p++) atomicAdd(&h[p][(k >> (dbits * p)) & dmask[p]], 1u);
                }
            }
        }
    }
    for (uint32_t i = nv * KPV + sub * 256 + tid; i < len; i += bpr * 256) {
        const uint32_t kraw = (uint32_t)kr[i];
        beyond |= (kraw - bias > kclamp) && kraw != (uint32_t)(KeyT)~(KeyT)0;
        const uint32_t k = min(kraw - bias, kclamp) >> begin_bit;
#pragma unroll
        for (int p = 0; p < PASSES; p++) atomicAdd(&h[p][(k >> (dbits * p)) & dmask[p]], 1u);
    }
    if (overflow && __any(beyond) && (threadIdx.x & 63) == 0) atomicAdd(overflow, 1u);
    GH_T(2);   // keys counted into LDS
    __syncthreads();
    GH_T(3);
#pragma unroll
    for (int p = 0; p < PASSES; p++)
        for (int d = tid; d < GS; d += 256)
            if (h[p][d]) atomicAdd(&ghist[(p * kOsRanges + x) * GS + d], h[p][d]);
#ifdef GSR_OS_TIMING
    GH_T(4);   // global adds issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GH_T(5);   // ... and acknowledged
    if (tid == 0 && blockIdx.x < 512) { ght[7] = ght_last; for (int q = 0; q < 8; q++) g_gh_dbg[blockIdx.x * 8 + q] = ght[q]; }
#endif
}

__device__ __forceinline__ void onesweep_rider_publish(const OsRider& rd, unsigned long long (*s_r)[16])
{
    unsigned long long a = 0ull, b = 0ull;
    // eight loads in flight per thread: this workgroup alone sums the producer's per-block shares (7 813 of them at 1 M Gaussians), the
    // plain loop ran one memory round trip per turn -- 31 in a row -- and the whole histogram launch waited for it (14 us where its
    // other workgroups take 6: tools/os_timing.sh)
    for (uint32_t q0 = threadIdx.x; q0 < rd.nparts; q0 += 8u * blockDim.x) {
        uint2 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t q = q0 + (uint32_t)u * blockDim.x;
            v[u] = q < rd.nparts ? rd.parts[q] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) { a += v[u].x; b += v[u].y; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); }
    if ((threadIdx.x & 63u) == 0u) { s_r[0][threadIdx.x >> 6] = a; s_r[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = b = 0ull;
        for (uint32_t w = 0; w < blockDim.x / 64u; w++) { a += s_r[0][w]; b += s_r[1][w]; }
        __atomic_store_n(rd.host + 2, (unsigned long long)g_onesweep_giveups, __ATOMIC_RELAXED);
        __atomic_store_n(rd.host + 3, b, __ATOMIC_RELAXED);
        __atomic_store_n(rd.host, a, __ATOMIC_RELAXED);
        __threadfence_system();
        __atomic_store_n(rd.host + 1, rd.seq, __ATOMIC_RELAXED);
        __threadfence_system();
    }
}

// 512 threads x 8 keys per tile: the same 4096-key tile as the three-kernel path, but half the ranking rounds per
// wave and twice the waves to hide the LDS / look-back latency behind (a workgroup's latency chain, not bandwidth,
// is what a pass costs)
// Measured (1 M depth keys / 4.5 M tile keys): 256x16 0.100 / 0.111 ms, 512x8 0.084 / 0.103, 1024x4 0.081 / 0.116,
// 512x4 (2048-key tiles) 0.086 / 0.127 -> the small latency-bound depth sort takes 1024 threads, the tile sort 512;
// 512x10 (5120-key tiles) for the tile sort: 0.096 -> 0.094 (its 4.5 M keys then fit the 1024 resident workgroups at once).
// One wave per 8 digits reading 64 predecessors per round trip (instead of one thread per digit reading 4): 3.7x SLOWER --
// the look-back is bound by the status traffic in L2, not by the length of the walk.
constexpr int kOsTile = 4096;

// NB = digit table size: 256, or 64 when the digits of the sort are at most 6 bits wide (a quarter of the LDS tables:
// five instead of three workgroups per CU for 16-bit keys).  Status words keep their 256-word stride in memory.
#ifdef GSR_OS_TIMING   // experiment build only (tools/os_timing.sh): where a workgroup of a sort pass spends its time
__device__ unsigned long long g_os_dbg[4096 * 8];   // [workgroup of the last u32 launch][phase]
#define OS_T(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) ost[k] += now_ - ost_last; ost_last = now_; } while (0)
#else
#define OS_T(k) do { } while (0)
#endif
template <typename KeyT, int NB>
__global__ __launch_bounds__(OsCfg<KeyT>::kThreads) void k_onesweep(const KeyT* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                         KeyT* __restrict__ kout, uint32_t* __restrict__ vout, uint32_t n, int shift,
                                                         const uint32_t* __restrict__ ghist /*[256] this pass*/,
                                                         uint32_t* __restrict__ status /*[nblocks][256]*/,
                                                         uint32_t* __restrict__ ticket,
                                                         const unsigned long long* __restrict__ n_dev, uint32_t dmask, int runs,
                                                         uint32_t bias = 0u, uint32_t kclamp = 0xffffffffu, OsRider rider = OsRider{})
{
    constexpr int GS = NB > 256 ? NB : 256;   // stride of the digit tables in memory (status words, per-run digit counts)
    constexpr int kOsThreads = OsCfg<KeyT>::kThreads, kOsTile = OsCfg<KeyT>::kTile, kOsIPT = kOsTile / kOsThreads, kOsWaves = kOsThreads / 64;
    if (rider.host && blockIdx.x == gridDim.x - 1) {   // (block-uniform; first pass of a sort whose producer counted the digits: no histogram kernel to ride on)
        __shared__ unsigned long long s_r[2][16];
        onesweep_rider_publish(rider, s_r);
    }
    if (n_dev) n = (uint32_t)min((unsigned long long)n, *n_dev);   // device-side count: tiles past it exit at once
    __shared__ unsigned long long s_mask[kOsWaves][NB];
    __shared__ uint32_t s_cnt[kOsWaves][NB];
    __shared__ KeyT s_keys[kOsTile];
    __shared__ uint32_t s_vals[kOsTile];
    __shared__ uint32_t s_start[NB];
    __shared__ uint32_t s_gbase[NB];
    __shared__ uint32_t s_wsum[kOsWaves];
    __shared__ uint32_t s_tile;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#ifdef GSR_OS_TIMING
    unsigned long long ost[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ost_last = __builtin_readcyclecounter();
    ost[6] = ost_last;
#endif
    // ticket == nullptr: the whole grid is co-resident (host checked), so blockIdx order is as good as ticket order and
    // the ~2 us global-atomic round trip at the head of every block's latency chain is saved
    // The FIRST pass cuts the key array into eight runs of tiles with their own look-back chains (run = blockIdx & 7): the
    // histogram kernel counted its digits per run, so a run's digit base is known up front.  Later passes see the keys in
    // the order the previous pass produced, for which no per-run counts exist: one chain (runs = 1).
    const uint32_t ntiles = (uint32_t)(((uint64_t)n + kOsTile - 1) / kOsTile);
    const uint32_t per = runs > 1 ? (ntiles + kOsRanges - 1) / kOsRanges : ntiles;
    const uint32_t run = runs > 1 ? (blockIdx.x & (kOsRanges - 1)) : 0u;
    uint32_t krun = runs > 1 ? blockIdx.x / kOsRanges : blockIdx.x;
    if (ticket) {
        if (tid == 0) s_tile = atomicAdd(ticket + run, 1u);
        __syncthreads();
        krun = s_tile;
    }
    const uint32_t run_start = run * per;
    const uint32_t tile = run_start + krun;
    if (krun >= per || tile >= ntiles) return;   // (uniform) the grid covers the capacity, the runs the device-side count
    const uint32_t base = tile * (uint32_t)kOsTile;
    const uint32_t valid = min((uint32_t)kOsTile, n - base);
    KeyT key[kOsIPT];
    uint32_t val[kOsIPT], dig[kOsIPT], rnk[kOsIPT];
#pragma unroll
    for (int r = 0; r < kOsIPT; r++) {
        const uint32_t p = (uint32_t)(wave * (kOsIPT * 64) + r * 64 + lane);   // wave-major: see wave_rank
        if (p < valid) {
            key[r] = kin[base + p];
            val[r] = vin[base + p];
            dig[r] = (min((uint32_t)key[r] - bias, kclamp) >> shift) & dmask;
        } else {
            key[r] = (KeyT)~(KeyT)0; val[r] = 0u; dig[r] = (uint32_t)(NB - 1);
        }
    }
    OS_T(0);   // keys requested (their first use is inside the ranking)
    wave_rank<kOsIPT, NB>(s_mask[wave], s_cnt[wave], dig, rnk, lane);
    __syncthreads();
    OS_T(1);   // loads landed + ranking + barrier
    // threads 0..255 = digits (the upper half of the block only takes part in the barriers of the scans)
    const bool is_digit = tid < NB;
    uint32_t mine = 0;   // this tile's count of digit `tid` (padding counted in digit 255; removed below)
    if (is_digit) {
#pragma unroll
        for (int w = 0; w < kOsWaves; w++) {
            const uint32_t c = s_cnt[w][tid];
            s_cnt[w][tid] = mine;
            mine += c;
        }
    }
    const uint32_t real = (tid == NB - 1) ? mine - ((uint32_t)kOsTile - valid) : mine;   // real keys of this digit
    uint32_t* my_status = status + (size_t)tile * GS + (tid & (NB - 1));
    if (is_digit) __hip_atomic_store(my_status, kOsLocal | real, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // local exclusive start of each digit (block scan of `mine`) and digit base (block scan of the global histogram)
    const uint32_t lstart = block_scan_excl<kOsWaves>(mine, s_wsum, tid);
    uint32_t gtot = 0u, gpre = 0u;   // keys of this digit in all runs / in the runs in front of this one
    if (is_digit) {
#pragma unroll
        for (int x = 0; x < kOsRanges; x++) {
            const uint32_t c = ghist[x * GS + tid];
            gtot += c;
            gpre += (uint32_t)x < run ? c : 0u;
        }
    }
    const uint32_t dbase = block_scan_excl<kOsWaves>(gtot, s_wsum, tid) + gpre;
    OS_T(2);   // publish + the two block scans
    if (is_digit) {
        // decoupled look-back over earlier tiles
        // (windows of 4 independent loads: the walk is a chain of L2 round trips, and the chain is what a block waits on)
        uint32_t excl = 0, polls = 0;
        const int t_lo = (int)run_start;   // the chain ends at the head of the run: what lies in front is in the digit base
        for (int t = (int)tile - 1; t >= t_lo;) {
            constexpr int kWin = 4;   // (8: same speed, 16: +20 % per pass -- the polls compete for the status lines in L2)
            uint32_t v[kWin];
#pragma unroll
            for (int q = 0; q < kWin; q++)
                v[q] = (t - q >= t_lo) ? __hip_atomic_load(status + (size_t)(t - q) * GS + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kOsIncl;
            bool done = false;
#pragma unroll
            for (int q = 0; q < kWin; q++) {
                if (done) break;
                const uint32_t f = v[q] & ~kOsMask;
                if (f == 0u) { done = true; break; }   // not published yet: re-poll from here
                excl += v[q] & kOsMask;
                t--;
                if (f == kOsIncl) { t = t_lo - 1; done = true; }
            }
            if (t >= t_lo && done) {
                __builtin_amdgcn_s_sleep(1);
                // a predecessor that never publishes (status words overwritten by a caller's bug) must not hang the device: after
                // ~10^7 polls -- seconds, where a healthy chain takes microseconds -- the walk gives up, the sort's result is
                // garbage, and g_onesweep_giveups says so (gsr_forward returns GSR_ERR_HIP)
                if (++polls > (1u << 23)) { atomicAdd(&g_onesweep_giveups, 1u); break; }
            }
        }
        __hip_atomic_store(my_status, kOsIncl | (excl + real), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_start[tid] = lstart;
        s_gbase[tid] = dbase + excl - lstart;
    }
    __syncthreads();
    OS_T(3);   // look-back (thread 0 is a digit thread) + barrier
#pragma unroll
    for (int r = 0; r < kOsIPT; r++) {
        const uint32_t lp = s_start[dig[r]] + s_cnt[wave][dig[r]] + rnk[r];
        s_keys[lp] = key[r];
        s_vals[lp] = val[r];
    }
    __syncthreads();
    OS_T(4);   // the tile in digit order in LDS
#pragma unroll
    for (int r = 0; r < kOsIPT; r++) {
        const uint32_t p = r * kOsThreads + tid;
        if (p < valid) {
            const KeyT k = s_keys[p];
            const uint32_t d = (min((uint32_t)k - bias, kclamp) >> shift) & dmask;
            const uint32_t g = s_gbase[d] + p;
            kout[g] = k;
            vout[g] = s_vals[p];
        }
    }
#ifdef GSR_OS_TIMING
    OS_T(5);
    if (tid == 0 && sizeof(KeyT) == 4 && blockIdx.x < 4096) { ost[7] = ost_last; for (int q = 0; q < 8; q++) g_os_dbg[blockIdx.x * 8 + q] = ost[q]; }
#endif
}

inline size_t onesweep_scratch_bytes(uint32_t n)
{
    const size_t nblocks = ((size_t)n + kOsTile - 1) / kOsTile;
    // ghist[4][runs][256] + tickets (padded) + status[4 passes][nblocks][256], or the 9-bit layout: ghist[3][runs][512] + tickets +
    // status[3][nblocks][512]
    const size_t nb = nblocks ? nblocks : 1;
    const size_t w8 = kOnesweepHeadWords + 4 * nb * 256, w9 = kOnesweepHeadWords9 + 3 * nb * 512;
    return ((w8 > w9 ? w8 : w9) * sizeof(uint32_t) + 255) & ~(size_t)255;
}

// begin_bit..end_bit in 8-bit passes (at most 4).  Same contract as radix_sort_pairs.  n_dev != nullptr: `n` is only a
// capacity (it sizes the grid and the scratch); the number of pairs is min(n, *n_dev), read on the device.
template <typename KeyT>
inline hipError_t onesweep_sort_pairs(KeyT* keys, uint32_t* vals, KeyT* keys_alt, uint32_t* vals_alt, uint32_t n, int begin_bit,
                                      int end_bit, void* scratch, int* in_alt, hipStream_t stream,
                                      const unsigned long long* n_dev = nullptr, bool head_prezeroed = false, bool hist_done = false,
                                      int max_digit_bits = 8, uint32_t bias = 0u, const OsRider* rider = nullptr)
{
    const bool wide_ = max_digit_bits == 9;
    const OsRider rd = (rider && (wide_ || !hist_done)) ? *rider : OsRider{};   // rides on the histogram launch ...
    const OsRider rd0 = (rider && !wide_ && hist_done) ? *rider : OsRider{};      // ... or, when the producer counted the digits, on the first pass
    *in_alt = 0;
    if (n == 0) return hipSuccess;
    const int bits = end_bit - begin_bit;
    const bool wide = max_digit_bits == 9;                    // 9-bit digits: 512-entry tables, table stride 512, own head layout
    const int passes = (bits + (wide ? 9 : 8) - 1) / (wide ? 9 : 8);
    if (passes < 1 || passes > 4 || (wide && (passes != 3 || sizeof(KeyT) != 4 || hist_done || begin_bit != 0))) return hipErrorInvalidValue;
    const uint32_t GS = wide ? 512u : 256u;
    const uint32_t kclamp = (wide && bits < 32) ? ((1u << bits) - 1u) : 0xffffffffu;
    if (!wide) bias = 0u;
    // balanced digits: 12 key bits sort as 6 + 6 rather than 8 + 4 (fewer same-digit collisions in the ranking, longer
    // runs per digit in the scatter); the last digit is narrower when the bits do not divide evenly
    const int dbits = (bits + passes - 1) / passes;   // (12 tile-key bits as 7 + 5 or 8 + 4: +2 / +5 us)  == onesweep_dbits(bits)
    const uint32_t nblocks = (n + OsCfg<KeyT>::kTile - 1) / OsCfg<KeyT>::kTile;   // (the scratch is sized for 4096-key tiles: never fewer words)
    uint32_t* ghist = static_cast<uint32_t*>(scratch);
    uint32_t* tickets = ghist + (wide ? 3 * kOsRanges * 512 : 4 * kOsRanges * 256);   // [pass][run]
    uint32_t* status = tickets + kOsTicketWords;
    // the head (histograms + tickets) must be zero before the histogram kernel; callers that can clear it in a kernel
    // of their own say so (kOnesweepHeadWordsMax words cover both layouts).  The status words are cleared by the histogram kernel itself.
    if (!head_prezeroed && !hist_done) {
        hipError_t e = hipMemsetAsync(scratch, 0, (wide ? kOnesweepHeadWords9 : kOnesweepHeadWords) * sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
    }
    const uint32_t status_words = (uint32_t)((size_t)passes * nblocks * GS);
    // (64 blocks of 1024 threads, to shorten the per-address chains of the closing global atomics: +5 us per sort)
    const uint32_t per_cap = (nblocks + kOsRanges - 1) / kOsRanges;
    const uint32_t hgrid = kOsRanges * (per_cap < 32u ? per_cap : 32u);   // histogram blocks: up to 32 per run
    // hist_done: the kernel that produced the keys counted the digits and cleared the status words (see onesweep_run_len)
    if (wide)
        hipLaunchKernelGGL((k_radix_ghist<KeyT, 3, 512>), dim3(hgrid), dim3(256), 0, stream, keys, n, begin_bit, ghist, n_dev, status, status_words, dbits, bits,
                           bias, kclamp, onesweep_overflow_word(scratch), rd);
    else if (!hist_done)
    switch (passes) {
        case 1: hipLaunchKernelGGL((k_radix_ghist<KeyT, 1>), dim3(hgrid), dim3(256), 0, stream, keys, n, begin_bit, ghist, n_dev, status, status_words, dbits, bits, 0u, 0xffffffffu, (unsigned int*)nullptr, rd); break;
        case 2: hipLaunchKernelGGL((k_radix_ghist<KeyT, 2>), dim3(hgrid), dim3(256), 0, stream, keys, n, begin_bit, ghist, n_dev, status, status_words, dbits, bits, 0u, 0xffffffffu, (unsigned int*)nullptr, rd); break;
        case 3: hipLaunchKernelGGL((k_radix_ghist<KeyT, 3>), dim3(hgrid), dim3(256), 0, stream, keys, n, begin_bit, ghist, n_dev, status, status_words, dbits, bits, 0u, 0xffffffffu, (unsigned int*)nullptr, rd); break;
        default: hipLaunchKernelGGL((k_radix_ghist<KeyT, 4>), dim3(hgrid), dim3(256), 0, stream, keys, n, begin_bit, ghist, n_dev, status, status_words, dbits, bits, 0u, 0xffffffffu, (unsigned int*)nullptr, rd); break;
    }
    KeyT *kin = keys, *kout = keys_alt;
    uint32_t *vin = vals, *vout = vals_alt;
    for (int p = 0; p < passes; p++) {
        const uint32_t pmask = (1u << (dbits < bits - dbits * p ? dbits : bits - dbits * p)) - 1u;
        const int runs = p == 0 ? kOsRanges : 1;
        const uint32_t pgrid = runs > 1 ? kOsRanges * per_cap : nblocks;
        const int wbits = dbits < bits - dbits * p ? dbits : bits - dbits * p;   // this pass's digit width
        const uint32_t resident = wbits <= 6 ? onesweep_resident_blocks(k_onesweep<KeyT, 64>, OsCfg<KeyT>::kThreads)
                                             : onesweep_resident_blocks(k_onesweep<KeyT, 256>, OsCfg<KeyT>::kThreads);
        uint32_t* tk_p = pgrid <= resident ? (uint32_t*)nullptr : tickets + p * kOsRanges;
        if (wide) {
            const uint32_t res9 = onesweep_resident_blocks(k_onesweep<KeyT, 512>, OsCfg<KeyT>::kThreads);
            uint32_t* tk9 = pgrid <= res9 ? (uint32_t*)nullptr : tickets + p * kOsRanges;
            hipLaunchKernelGGL((k_onesweep<KeyT, 512>), dim3(pgrid), dim3(OsCfg<KeyT>::kThreads), 0, stream, kin, vin, kout, vout, n,
                               begin_bit + dbits * p, ghist + p * kOsRanges * 512, status + (size_t)p * nblocks * 512, tk9, n_dev, pmask, runs, bias, kclamp);
        } else if (wbits <= 6)
            hipLaunchKernelGGL((k_onesweep<KeyT, 64>), dim3(pgrid), dim3(OsCfg<KeyT>::kThreads), 0, stream, kin, vin, kout, vout, n,
                               begin_bit + dbits * p, ghist + p * kOsRanges * 256, status + (size_t)p * nblocks * 256, tk_p, n_dev, pmask, runs,
                               0u, 0xffffffffu, p == 0 ? rd0 : OsRider{});
        else
            hipLaunchKernelGGL((k_onesweep<KeyT, 256>), dim3(pgrid), dim3(OsCfg<KeyT>::kThreads), 0, stream, kin, vin, kout, vout, n,
                               begin_bit + dbits * p, ghist + p * kOsRanges * 256, status + (size_t)p * nblocks * 256, tk_p, n_dev, pmask, runs,
                               0u, 0xffffffffu, p == 0 ? rd0 : OsRider{});
        KeyT* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
        *in_alt ^= 1;
    }
    return hipGetLastError();
}

inline size_t radix_scratch_bytes(uint32_t n)
{
    const size_t nblocks = ((size_t)n + kSortTile - 1) / kSortTile;
    const size_t osblocks = ((size_t)n + kOsTile - 1) / kOsTile;
    const size_t three_kernel = ((256 * (nblocks ? nblocks : 1) + 8 * 256) * sizeof(uint32_t) + 255) & ~(size_t)255;   // hist + totals
    const size_t onesweep = onesweep_scratch_bytes(n);   // (either digit-table layout)
    (void)osblocks;
    return three_kernel > onesweep ? three_kernel : onesweep;
}

// Sorts bits [begin_bit, end_bit) in 8-bit passes, ping-ponging between (keys,vals) and (keys_alt,vals_alt).
// Returns 1 in *in_alt if the result ended in the alt buffers.
template <typename KeyT>
inline hipError_t radix_sort_pairs(KeyT* keys, uint32_t* vals, KeyT* keys_alt, uint32_t* vals_alt, uint32_t n, int begin_bit,
                                   int end_bit, void* scratch, int* in_alt, hipStream_t stream)
{
    *in_alt = 0;
    if (n == 0) return hipSuccess;
    const uint32_t nblocks = (n + kSortTile - 1) / kSortTile;
    uint32_t* hist = static_cast<uint32_t*>(scratch);
    uint32_t* totals = hist + (size_t)256 * nblocks;
    hipError_t e = hipMemsetAsync(totals, 0, 8 * 256 * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    KeyT *kin = keys, *kout = keys_alt;
    uint32_t *vin = vals, *vout = vals_alt;
    for (int shift = begin_bit; shift < end_bit; shift += 8, totals += 256) {
        hipLaunchKernelGGL(k_radix_hist<KeyT>, dim3(nblocks), dim3(kSortThreads), 0, stream, kin, n, shift, hist, nblocks, totals);
        hipLaunchKernelGGL(k_radix_scan, dim3(256), dim3(256), 0, stream, hist, nblocks, totals);
        hipLaunchKernelGGL(k_radix_scatter<KeyT>, dim3(nblocks), dim3(kSortThreads), 0, stream, kin, vin, kout, vout, n, shift,
                           hist, nblocks);
        KeyT* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
        *in_alt ^= 1;
    }
    return hipGetLastError();
}

}  // namespace gsr
