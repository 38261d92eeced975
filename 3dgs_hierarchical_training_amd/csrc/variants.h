// variants.h -- launchers of the A/B blend kernels in variants.hip (compiled only with -DGSR_AB_VARIANTS)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gsr_math.h"

namespace gsr {

// ppt: 1 / 3 / 4 = one workgroup per tile with 1 / 2 / 4 pixels per lane, 2 = packed two-pixel kernel, 5 = one wave per 8x8
// sub-tile with a lane mask for finished pixels (the predecessor of the default kernel).  Returns false for an unknown ppt.
bool launch_blend_fwd_variant(int ppt, int W, int H, int tiles_x, int T, const uint2* ranges, const uint32_t* list, const Splat* splat,
                              const float* bg, float* out_color, float* out_depth, float* out_alpha, float* img, uint32_t* staged,
                              int tile_map, float* ckpt, int ckpt_first, hipStream_t st);
// ppt: 1 / 3 / 4 = scalar kernels with 1 / 2 / 4 pixels per lane
bool launch_blend_bwd_variant(int ppt, int W, int H, int tiles_x, int T, const uint2* ranges, const uint32_t* list, const Splat* splat,
                              const float* bg, const float* img, const float* g_color, const float* g_depth, const float* g_alpha,
                              float* ggrad, hipStream_t st);

}  // namespace gsr
