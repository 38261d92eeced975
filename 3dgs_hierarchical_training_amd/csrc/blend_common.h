// blend_common.h -- what the blend kernels of gsr_kernels.hip share: image-state layout,
// checkpoint layout, tile -> XCD maps, the wave64 DPP sum and the packed-float helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gsr_math.h"

namespace gsr {

// wave64 sum via DPP (row_shr 1,2,4,8 then row_bcast15 / row_bcast31): total lands in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v)
{
    const int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(r);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v = dpp_add<0x111, 0xf>(v);   // row_shr:1
    v = dpp_add<0x112, 0xf>(v);   // row_shr:2
    v = dpp_add<0x114, 0xf>(v);   // row_shr:4
    v = dpp_add<0x118, 0xf>(v);   // row_shr:8
    v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 -> rows 1,3
    v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 -> rows 2,3
    return v;
}

// image state planes (floats): 0 final_T, 1 n_contrib(u32), 2..4 C, 5 D, 6 A
constexpr int kImgPlanes = 7;
// The backward of a tile is split over several workgroups: at every 128-instance boundary of a tile's list (from batch
// `first` on; first = 1 by default, i.e. every boundary) the forward leaves a checkpoint of every pixel's running state
// (T, C.rgb, D, A), so a workgroup can start its front-to-back replay there instead of at instance 0 -- (tile, batch)
// pieces are independent, which both balances long lists and puts three to four times more waves in flight on a
// 2 170-tile frame.  Slot of (tile, batch k >= first): (ranges[tile].x >> 7) + tile + k - first (non-overlapping:
// floor(a) + floor(b) + 1 <= floor(a + b) + 1).
constexpr int kCkptPlanes = 6, kCkptFloats = kCkptPlanes * kTile * kTile;

__device__ __forceinline__ int xcd_tile(int b, int T)
{
    const int per = (T + 7) >> 3;
    return (b & 7) * per + (b >> 3);
}

// tile handled by slot `s` of XCD `x` (0..7).  map 0 "banded": XCD x owns the contiguous tiles [x per, (x+1) per);
// map 1 "interleaved": tile t lives on XCD t % 8; map 2 "block-interleaved": 2x2 blocks of tiles are dealt round-robin to
// the XCDs (block j on XCD j % 8), which still spreads a spatially coherent hot region over all XCDs but keeps most of
// the tiles a splat touches behind one L2.  Returns -1 for a slot beyond the image.
__device__ __forceinline__ int slot_tile(int map, int x, int s, int T, int tiles_x)
{
    if (map == 0) { const int per = (T + 7) >> 3; const int t = x * per + s; return (s < per && t < T) ? t : -1; }
    if (map == 1) { const int t = s * 8 + x; return t < T ? t : -1; }
    const int tiles_y = T / tiles_x, bx_n = (tiles_x + 1) >> 1, by_n = (tiles_y + 1) >> 1;
    const int j = (s >> 2) * 8 + x, w = s & 3;
    if (j >= bx_n * by_n) return -1;
    const int tx = 2 * (j % bx_n) + (w & 1), ty = 2 * (j / bx_n) + (w >> 1);
    return (tx < tiles_x && ty < tiles_y) ? ty * tiles_x + tx : -1;
}
// slots per XCD that cover every tile under `map`
static inline int slots_per_xcd(int map, int T, int tiles_x)
{
    if (map != 2) return (T + 7) / 8;
    const int tiles_y = T / tiles_x, nblk = ((tiles_x + 1) / 2) * ((tiles_y + 1) / 2);
    return 4 * ((nblk + 7) / 8);
}

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c)   // per-component fused multiply-add (v_pk_fma_f32)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_elementwise_fma(a, b, c);
#else
    return f2{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)};
#endif
}

// ggrad record (12 floats / Gaussian): gx gy gA gB gC gop gr gg gb gz - -
constexpr int kGG = 12;
constexpr int kDetStride = 10;   // floats per (tile, instance) slot of the deterministic backward

}  // namespace gsr
