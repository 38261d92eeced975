/*
 * include/gsr.h -- C ABI of the MI355X (gfx950) Gaussian rasterizer library libgsr_hip.so.
 *
 * Drop-in boundary.  The reference binds its rasterizer through the Python module
 * `diff_gaussian_rasterization` (call sites: /root/reference/scene/gaussian_model_ht.py:809-880,
 * /root/reference/gaussian_renderer/__init__.py:38-96, /root/reference/scene/gaussian_model.py:935-1009),
 * whose native extension `_C` is an UN-VENDORED submodule (/root/reference/.gitmodules:4-6).  The three entry
 * points a maintainer's FFI for this path would bind are restated here as plain C:
 *
 *   gsr_forward       <- the extension's forward  ("rasterize_gaussians"), invoked by GaussianRasterizer.forward
 *                        as called at gaussian_model_ht.py:871-880; returns what :881-894 unpacks
 *   gsr_backward      <- the extension's backward ("rasterize_gaussians_backward"), reached by loss.backward()
 *                        at /root/reference/trainer/ht3dgs_trainer.py:135; produces the grads consumed at
 *                        gaussian_model_ht.py:718-721 (means2D) and by the Adam groups :275-289
 *   gsr_mark_visible  <- the extension's "mark_visible" (GaussianRasterizer.markVisible; unused by the reference)
 *
 * Rules: extern "C", raw DEVICE pointers and sizes only, no torch types, no exceptions.  Every call is
 * asynchronous on `stream` (a hipStream_t passed as void*) except for one device->host read of the instance
 * count inside gsr_forward.  Return 0 on success, a negative GSR_ERR_* otherwise; gsr_last_error() gives text.
 * All float tensors are binary32, contiguous.  4x4 matrices are read linearly as the reference stores them
 * (transposed, /root/reference/scene/cameras.py:76-98).
 */
#ifndef GSR_H
#define GSR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_OK 0
#define GSR_ERR_ARG (-1)
#define GSR_ERR_HIP (-2)
#define GSR_ERR_ALLOC (-3)
#define GSR_ERR_RANGE (-4)

#define GSR_ALLOC_BINNING 0 /* kept by the caller until gsr_backward */
#define GSR_ALLOC_SCRATCH 1 /* may be released (stream-ordered) as soon as gsr_forward returns */

/* Allocator callback: return a device pointer to `bytes` bytes (>=256-byte aligned) or NULL. */
typedef void* (*gsr_alloc_fn)(size_t bytes, int tag, void* user);

/* Batched rendering of INDEPENDENT models in one launch chain (stage A of the reference fits F - 1 single-image models that share
 * nothing, /root/reference/trainer/ht3dgs_trainer.py:697-698, :336-431; each is a chain of ~17 dependent kernels that fill a
 * fraction of the chip).  B models live in ONE parameter store of N Gaussians; model b owns the 128-Gaussian blocks
 * [first_block[b], first_block[b + 1]) -- every model is padded to a multiple of 128 Gaussians (padding = Gaussians that are culled,
 * e.g. opacity logit -30) and N = 128 first_block[B].  With a batch attached:
 *   viewmatrix / projmatrix / campos / points_transform (and those of GsrNextView) are arrays of B entries (16 / 16 / 3 / 12 floats);
 *   out_color is [B,3,H,W], out_depth / out_alpha [B,1,H,W], the upstream gradients of gsr_backward likewise;
 *   d_viewmatrix / d_projmatrix / d_campos / d_points_transform are arrays of B entries;
 *   `image` holds gsr_image_bytes_batched(W, H, B) bytes;  W, H, tanfov, bg, sh_degree, scale_modifier are shared.
 * Every model's image, radii and gradients are bit-identical with rendering it alone: the B images form one tall tile grid
 * (tile id = b T + tile), a tile's list only holds its own model's Gaussians, and one depth sort orders them all.
 * first_block is a HOST pointer (B + 1 entries).  B <= 16.  NULL / B <= 1 = the ordinary single-model call. */
typedef struct GsrBatch {
    int32_t B;
    const int32_t* first_block;
} GsrBatch;

typedef struct GsrForwardArgs {
    int32_t N;           /* Gaussians */
    int32_t M;           /* SH coefficients stored per Gaussian (row stride of shs), e.g. 16 */
    int32_t D;           /* active SH degree 0..3 (raster_settings.sh_degree) */
    int32_t W, H;        /* image_width, image_height */
    int32_t prefiltered; /* accepted, ignored (reference always passes False: gaussian_model_ht.py:820) */
    int32_t debug;       /* accepted, ignored (reference always passes False: gaussian_model_ht.py:821) */
    float scale_modifier, tanfovx, tanfovy;
    const float* means3D;        /* [N,3] */
    const float* scales;         /* [N,3]  or NULL when cov3D_precomp is given */
    const float* rotations;      /* [N,4]  (w,x,y,z) */
    const float* cov3D_precomp;  /* [N,6]  xx,xy,xz,yy,yz,zz or NULL */
    const float* opacities;      /* [N] */
    const float* shs;            /* [N,M,3] or NULL when colors_precomp is given */
    const float* colors_precomp; /* [N,3] or NULL */
    const float* viewmatrix;     /* 16 floats, device */
    const float* projmatrix;     /* 16 floats, device */
    const float* campos;         /* 3 floats, device */
    const float* bg;             /* 3 floats, device */
    float* out_color;            /* [3,H,W] */
    float* out_depth;            /* [1,H,W] */
    float* out_alpha;            /* [1,H,W] */
    int32_t* radii;              /* [N] */
    void* geom;                  /* gsr_geom_bytes(N) bytes, kept until backward */
    void* image;                 /* gsr_image_bytes(W,H) bytes, kept until backward */
    gsr_alloc_fn alloc;          /* called for the R-sized buffers once R is known */
    void* alloc_user;
    /* ---- extension, "next" row f-2 (fused activations; zero/NULL = the reference's calling convention) ---- */
    const float* shs_rest;       /* if non-NULL: shs is _features_dc [N,1,3] and shs_rest is _features_rest [N,M-1,3]
                                    (gaussian_model_ht.py:176-179 concatenates them every call) */
    int32_t raw_params;          /* 1: scales = log-scales, rotations un-normalised, opacities = logits; the
                                    activations of gaussian_model_ht.py:49-65,128-133,187-188 run in-kernel */
    /* ---- extension, "next" row f-4: fused pose action.  NULL = none.  12 floats, row-major 3x4 [R | t], device.
     * Every mean is replaced by R p + t before anything else is computed -- the in-kernel form of
     * `xyz = self.P[k].retr().act(self._xyz.clone())` (gaussian_model_ht.py:135-148): positions move, the Gaussians'
     * own rotations / scales / SH do not, exactly as the reference's get_xyz has it. */
    const float* points_transform;
    /* ---- "prepare in backward" (see GsrNextView): the buffer a preceding gsr_backward filled through
     * GsrBackwardArgs::prepared_out for THIS camera and THESE parameter values.  Then `geom` must be the same pointer (the
     * splat records are its first gsr_geom_bytes(N) bytes), k_preprocess is skipped, and the result is bit-identical to a
     * forward without it.  NULL = ordinary forward.  The buffer is consumed: it also holds the depth sort's scratch, pre-cleared
     * (and for N <= 262 144 with the digit counts already in it) by that backward, and its sort keys are sorted in place --
     * one forward per hand-over. */
    void* prepared;
    const struct GsrBatch* batch; /* NULL = one model (see GsrBatch) */
    /* ---- round 5: the caller's id of the FRAME this render shows (the reference's `viewpoint_camera.uid`, trainer/trainer.py:573;
     * any non-zero number that is the same whenever the same frame is rendered).  0 = none.  Speed only, never the result: the
     * forward blend places its waves by the work each did at the previous render of the same frame ("blend_balance"); with an id
     * that frame is recognised whatever its pose does in between, without one it is recognised by its pose (view matrix AND
     * points_transform within "view_pose_tol_e6" of the previous render's -- the reference keeps an identity camera and moves the
     * points, gaussian_model_ht.py:135-148, and steps the pose after every render, ht3dgs_trainer.py:162-166). */
    int64_t view_id;
    /* ---- round 5: two results the reference's render wrapper derives with a torch launch each (gaussian_model_ht.py:883, :905), written
     * by kernels that hold the values anyway.  NULL = not wanted.
     *   out_color_clamped [3,H,W] (a batch: [B,3,H,W]): clamp(out_color, 0, 1), by the forward blend;
     *   visible [N] bytes: radii > 0, by the preprocess -- NOT written by a forward that takes a `prepared` buffer (no preprocess runs). */
    float* out_color_clamped;
    uint8_t* visible;
} GsrForwardArgs;

typedef struct GsrForwardOut {
    int64_t num_rendered; /* R: (tile, Gaussian) instances */
    void* binning;        /* pointer returned by alloc(GSR_ALLOC_BINNING); pass to gsr_backward */
    size_t binning_bytes;
    int64_t binning_capacity; /* instances the binning buffer was laid out for (>= R when the forward ran speculatively);
                                 pass to gsr_backward together with `binning` */
    int64_t forward_flags;    /* what this forward fixed for its backward (blend kernel variant, tile -> XCD map, checkpoint
                                 layout): pass to gsr_backward unchanged.  gsr_set_option calls between the two then cannot
                                 make the backward read checkpoints the forward never wrote */
} GsrForwardOut;

/* Optimizer-in-backward: with raw_params = 1, shs (= _features_dc) + shs_rest given and this struct attached,
 * gsr_backward applies the Adam step of /root/reference/scene/gaussian_model_ht.py:275-289 (torch.optim.Adam, no
 * amsgrad / weight decay; ht3dgs_trainer.py:159-166 calls step() right after backward()) to the six parameter tensors
 * IN PLACE inside the per-Gaussian backward kernel, instead of writing their gradients: the gradient never makes
 * the round trip through HBM (2180 -> ~1480 bytes per Gaussian for backward + optimizer).  The parameter pointers of
 * GsrBackwardArgs (means3D, shs, shs_rest, opacities, scales, rotations) are then written through; d_means3D,
 * d_opacities, d_shs, d_shs_rest, d_scales, d_rotations are ignored (may be NULL); d_means2D is still produced
 * (densification statistics, gaussian_model_ht.py:718-721).  Group order of lr / exp_avg / exp_avg_sq:
 * 0 xyz, 1 f_dc, 2 f_rest, 3 opacity, 4 scaling, 5 rotation.  `step` is the 1-based step count.
 * exp_avg[2] == exp_avg_sq[2] == NULL: the f_rest group is left alone -- no read or write of its 45 floats of parameters and 90 of
 * moments per Gaussian, three quarters of the update's traffic.  Accepted only for a render at sh_degree 0 that prepares no view
 * of a higher degree (the group's gradient is then identically zero), and CORRECT only if the caller knows the group's moments
 * to be all zero: Adam with g = m = v = 0 leaves parameter and moments unchanged bit for bit (0 / (0 + eps) = 0), so skipping it
 * is the same update.  That is a model's state from its creation until its first step at degree >= 1 -- all of stage A and the
 * first 1 000 iterations of every leaf (gaussian_model_ht.py:68, :193-195); optim.FusedAdam tracks it. */
typedef struct GsrFusedAdam {
    float beta1, beta2, eps;
    int32_t reserved;
    int64_t step;
    float lr[6];
    float* exp_avg[6];
    float* exp_avg_sq[6];
    int32_t step_lag[6];    /* round 4: group q is at step `step - step_lag[q]` (its bias corrections use that count); all zero =
                             * the six groups in lockstep.  The reference drops a group's update when its tensor is replaced
                             * between backward() and optimizer.step() (opacity reset: gaussian_model_ht.py:468-474,
                             * ht3dgs_trainer.py:153-160), after which that group's count stays behind the others'. */
    /* round 4, DEFERRED application: when a group's three pointers are set, its updated parameter rows and moments are written
     * THERE instead of in place (inputs untouched), and the caller adopts them -- swaps the buffers -- when its `optimizer.step()`
     * runs, or drops them when the step never comes (the reference's trainer densifies between backward() and step() and then
     * steps nothing: ht3dgs_trainer.py:137-160).  All NULL = in place.  Not together with next_view. */
    float* param_out[6];
    float* exp_avg_out[6];
    float* exp_avg_sq_out[6];
} GsrFusedAdam;

/* "Prepare in backward" (extension f-2, with fused_adam only).  Training renders the same parameters again right after
 * their update; when the caller knows the NEXT camera at backward time, the per-Gaussian backward kernel -- which holds
 * each Gaussian's freshly updated parameters in registers / LDS -- also runs the forward preprocess of that next render
 * (projection, tile test, SH colour, sort keys) and leaves the result in `prepared_out`; the next gsr_forward takes it
 * through GsrForwardArgs::prepared and skips its preprocess kernel: no second read of the 236 bytes per Gaussian, and
 * the ~2 400 VALU instructions per wave of the preprocess hide under the HBM time of the backward kernel.
 * Needs raw_params, shs + shs_rest with M = 16 stored coefficients at any active degree D = 0..3 (gsr_prepare_supported) --
 * the reference's models store 16 coefficients from the start and raise the active degree once per 1 000 iterations
 * (/root/reference/scene/gaussian_model_ht.py:68,193-195); GsrNextView::D is this render's degree or one above it (an
 * `oneupSHdegree` between the two steps keeps the hand-over).  The caller guarantees that the parameters are not modified
 * between this backward and that forward. */
typedef struct GsrNextView {
    int32_t W, H, D;
    float scale_modifier, tanfovx, tanfovy;
    const float *viewmatrix, *projmatrix, *campos; /* device, as in GsrForwardArgs */
    const float* points_transform;                  /* device, 12 floats, or NULL */
} GsrNextView;

/* Per-iteration densification statistics, folded into the per-Gaussian backward kernel (it holds the screen-space gradient in
 * registers).  What the reference's train_step does after every backward while iteration < densify_until_iter
 * (/root/reference/trainer/ht3dgs_trainer.py:137-147, /root/reference/scene/gaussian_model_ht.py:718-721), for the visible
 * Gaussians (radii > 0):  max_radii2D = max(max_radii2D, radii);  xyz_gradient_accum += |d_means2D[:, :2]|;  denom += 1.
 * radii = the forward's int32 output; the three float arrays hold N elements each and are updated in place. */
typedef struct GsrDensifyStats {
    const int32_t* radii;
    float* xyz_gradient_accum;
    float* denom;
    float* max_radii2D;
} GsrDensifyStats;

typedef struct GsrBackwardArgs {
    int32_t N, M, D, W, H;
    float scale_modifier, tanfovx, tanfovy;
    const float *means3D, *scales, *rotations, *cov3D_precomp, *opacities, *shs, *colors_precomp;
    const float *viewmatrix, *projmatrix, *campos, *bg;
    const void* geom;    /* from forward */
    const void* image;   /* from forward */
    void* binning;       /* from forward.  WRITTEN by gsr_backward (round 4): the backward blend's work-item lists and queue heads live in
                          * this buffer, so two backwards over the SAME forward output must not run concurrently (serialise them on one
                          * stream); a repeated backward on one stream is fine -- the lists are rebuilt by every call */
    int64_t num_rendered;
    const float* grad_color; /* [3,H,W] or NULL */
    const float* grad_depth; /* [1,H,W] or NULL */
    const float* grad_alpha; /* [1,H,W] or NULL */
    float* d_means3D;        /* [N,3] */
    float* d_means2D;        /* [N,3]  (d/d ndc x, d/d ndc y, 0): gaussian_model_ht.py:718-721 */
    float* d_opacities;      /* [N] */
    float* d_colors_precomp; /* [N,3] or NULL */
    float* d_shs;            /* [N,M,3] or NULL */
    float* d_scales;         /* [N,3] or NULL */
    float* d_rotations;      /* [N,4] or NULL */
    float* d_cov3D_precomp;  /* [N,6] or NULL */
    void* scratch;           /* gsr_backward_scratch_bytes(N) bytes */
    /* ---- extension f-2: same meaning as in GsrForwardArgs; gradients are returned w.r.t. the raw parameters ---- */
    const float* shs_rest;
    float* d_shs_rest;       /* [N,M-1,3] when shs_rest is given (then d_shs is [N,1,3]) */
    int32_t raw_params;
    /* ---- camera gradients (north_star: dL/dviewmatrix; BASELINE config 5).  NULL = not wanted.  Entries follow
     * the linear storage of the inputs.  projmatrix row 2 (clip z) does not influence the render: its grad is 0. */
    float* d_viewmatrix;     /* 16 */
    float* d_projmatrix;     /* 16 */
    float* d_campos;         /* 3 */
    /* ---- optimizer-in-backward (extension f-2, see GsrFusedAdam below).  NULL = plain backward. */
    const struct GsrFusedAdam* fused_adam;
    /* ---- fused pose action (f-4): the transform given to the forward, and where dL/d(transform) (12 floats, same
     * layout) goes; d_means3D is then the gradient w.r.t. the UNtransformed means (R^T dL/dp'). */
    const float* points_transform;
    float* d_points_transform;
    int64_t binning_capacity; /* GsrForwardOut::binning_capacity of the forward (0 = num_rendered) */
    int64_t forward_flags;    /* GsrForwardOut::forward_flags of the forward (0 = round-1 caller: process-wide options) */
    const struct GsrNextView* next_view; /* NULL = none; otherwise prepared_out must point at gsr_prepared_bytes(N) bytes */
    void* prepared_out;
    const struct GsrDensifyStats* densify_stats; /* NULL = none */
    const struct GsrBatch* batch;                 /* the forward's batch (NULL = one model) */
} GsrBackwardArgs;

size_t gsr_geom_bytes(int32_t N);
size_t gsr_image_bytes(int32_t W, int32_t H);
size_t gsr_image_bytes_batched(int32_t W, int32_t H, int32_t B); /* image workspace of a batched render (GsrBatch) */
/* byte offset inside the image workspace of the per-tile uint32 "instances actually staged" counters that
 * the forward blend writes (their sum is R_eff of the roofline accounting) */
size_t gsr_image_staged_offset(int32_t W, int32_t H);
size_t gsr_forward_scratch_bytes(int32_t N);
size_t gsr_binning_bytes(int64_t R, int32_t W, int32_t H);
/* upper bound of what the forward asks its allocator for (GSR_ALLOC_SCRATCH): sized for 32-bit tile keys, which images
 * above 65 536 tiles use; smaller images carry 16-bit keys and ask for 4 R bytes less */
size_t gsr_binning_scratch_bytes(int64_t R);
size_t gsr_backward_scratch_bytes(int32_t N);
size_t gsr_prepared_bytes(int32_t N);                       /* "prepare in backward": size of the hand-over buffer */
size_t gsr_prepared_radii_offset(int32_t N);                /* where its int32 radii[N] start: a forward given `radii` = that
                                                               address uses them in place instead of copying them out */
int gsr_prepare_supported(int32_t M, int32_t D, int32_t raw_params); /* 1 when gsr_backward can prepare the next view */

int gsr_forward(const GsrForwardArgs* args, GsrForwardOut* out, void* stream);
int gsr_backward(const GsrBackwardArgs* args, void* stream);
int gsr_mark_visible(int32_t N, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, void* stream);

const char* gsr_last_error(void);
/* 100 * major + minor.  110 (round 6): GsrForwardArgs ends with view_id, out_color_clamped, visible (appended in round 5 under
 * version 100: a caller compiled against a shorter struct must be rebuilt -- check gsr_version() >= 110 AND
 * gsr_struct_bytes(0) == sizeof(GsrForwardArgs), gsr_struct_bytes(1) == sizeof(GsrBackwardArgs) at start-up). */
int gsr_version(void);
size_t gsr_struct_bytes(int32_t which); /* 0 GsrForwardArgs, 1 GsrBackwardArgs, 2 GsrForwardOut; anything else 0 */

/* Process-wide knobs (value 0 = default unless stated):
 *   "blend_fwd_ppt"        forward blend kernel: 7 = one wave per 8x8 sub-tile, finished pixels encoded in the sign of T,
 *                          instances the sub-tile cannot see skipped by reach bits (default); 6 = without the reach bits.
 *                          Any other value is refused (the A/B kernels rounds 1-4 selected with 1-5 left the tree in round 5)
 *   "blend_bwd_ppt"        backward blend kernel: 2 (default; 0 = default) = two pixels per lane, packed math, two waves per tile; 1 = one pixel per lane,
 *                          four waves per tile, four reach bits per staged instance (k_blend_bwd1: same results within rounding; -7 % of the kernel
 *                          on random scenes of up to ~130 k Gaussians, +2 % on pixel-sized splats, equal from 300 k on)
 *   "ab_variants"          query kept for callers of the round-4 ABI: always returns 0 (no A/B kernels in the library)
 *   "sort_algo"            2 = onesweep for both sorts (default); 1 = onesweep depth sort only; 0 = hist + scan + scatter
 *   "bwd_split"            1 = the backward of a tile is ONE work item (off); anything else (default 0) = one item per 128-instance
 *                          batch beyond the first "ckpt_first" (default 1) batches, each resuming from the per-pixel checkpoint the
 *                          forward leaves at every 128-instance boundary.  (Round 4: the items are listed by k_bwd_prologue and
 *                          replayed by persistent workgroups; up to round 3 the value was the number of workgroups per tile.)
 *   "depth_sort9"          1 (default) = depth sort of models above "prep_hist_max_n" Gaussians in THREE 9-bit passes over the 27 bits
 *                          of (key - bits(0.2f)): depths between the near plane and 13 107 units; a visible Gaussian beyond that
 *                          makes the forward sort again on all 32 bits (counter "depth_window_resorts") and keeps that caller on
 *                          the four 8-bit passes; 0 = always four 8-bit passes.  Same order either way
 *   "direct_binning"       1 (default) = the per-tile lists are built by direct placement -- per-chunk tile histograms over the
 *                          depth-ordered Gaussians, column scans, one wave per chunk that places every (Gaussian, tile) pair at
 *                          its final position -- on frames of up to 4 096 tiles; 0 = emit + stable tile sort + ranges (what larger
 *                          frames always take).  Same list, bit for bit
 *   "tile_sort"            1 (default) = where the direct binning runs on a single render AND the caller's tile lists are short (at most
 *                          "tile_sort_max_avg" pairs per tile on average, by the caller's last instance count), the global depth sort is
 *                          left out: the binning walks the Gaussians in index order, places (depth key, Gaussian) pairs, and one
 *                          workgroup per tile sorts its segment by key, stable; 2 = wherever the direct binning runs; 0 = never.
 *                          Same list, bit for bit (equal keys stay in index order, as under the stable global sort)
 *   "tile_sort_max_avg"    (default 700) see "tile_sort"
 *   "blend_balance"        1 (default) = the forward blend places its sub-tile waves by the visits each took at the previous
 *                          render of the same frame (device-side cache of 128 frames per frame size, least recently used out; a
 *                          frame is recognised by GsrForwardArgs::view_id or, without one, by its pose; single renders through
 *                          the default kernel); 0 = dispatch order = tile order.  Same image either way
 *   "direct_slab_tiles"    (default 0 = off) frames above 4 096 tiles on the direct route: the tile grid is cut into slabs of whole tile
 *                          rows of at most this many tiles and a chunk is walked by one wave per slab.  Same list bit for bit; measured
 *                          no faster than the sort route such frames take by default
 *   "small_sort9"          (default 1) smallest model for which a forward that runs its own preprocess sorts its depth keys in three
 *                          9-bit passes over the 27-bit window instead of four 8-bit ones (it launches a digit histogram either way);
 *                          0 = only models above 262 144 Gaussians.  Same order either way (a depth beyond the window is detected
 *                          and sorted again on all 32 bits)
 *   "list_cut"             1 (default) = for models of at least "list_cut_min_n" Gaussians (default 2 000 000: where it was measured to pay;
 *                          2 = whatever the size), on the plain direct-binning route (frames of up to 4 096 tiles behind the global depth sort,
 *                          single renders, default blend, "blend_balance" on) the tile lists are WRITTEN only up to where the tiles
 *                          stopped at the previous render of the same frame (the frame is recognised as for "blend_balance"; a tile
 *                          remembers the view depth of the last instance any of its four waves staged, x (1 + "list_cut_margin_e3" /
 *                          1000), rounded up to a chunk of the depth order).  One cut for the frame -- the chunk behind which at
 *                          most "list_cut_deep" tiles' cuts lie; the chunks up to it are scattered as ever, for every tile -- and
 *                          the few deeper tiles are served one by one in the chunks behind it.  Counts, scans, tile bases, R and
 *                          every pair's position are the full binning's; the blends get the end of the valid prefix as the range
 *                          end.  Verified and repaired on the device, never by the host: a wave that runs out of a cut list with a
 *                          live pixel flags its tile, and two launches that follow every such render -- the flagged tiles' left-out
 *                          pairs, the blend over the flagged tiles' full lists -- find nothing to do otherwise.  Image, radii,
 *                          every pixel's contributors, checkpoints and all gradients are bit-identical to full lists; ranges[t].y
 *                          and the list words behind it are what differ (gsr_debug_read_binning returns the valid prefixes).
 *                          0 = full lists (gsr_debug_list_cut_stats)
 *   "list_cut_margin_e3"   (default 50), "list_cut_deep" (default 8): see "list_cut"
 *   "early_r"              1 (default) = the host learns the instance count from the preprocess's per-block sums, published by
 *                          the depth sort's first kernel, instead of from the scan behind sort + tile counts; 0 = from the scan.
 *                          The scan still reports its own total and the sorts' give-up counter into spare words of the pinned slot;
 *                          the next gsr_forward that takes the slot (waiting for the report if it must) and every gsr_backward (if
 *                          it has arrived) hold the early values against them and FAIL (GSR_ERR_HIP) on a difference: a forward
 *                          whose count or sort went wrong is reported by the next call into the library, not never
 *                          (counters "late_checks", "late_mismatches"; "debug_late_bias": tests, falsifies the next remembered count)
 *   "view_pose_tol_e6"     (default 2000 = 2e-3) without a view id a render belongs to the cached frame whose view matrix and
 *                          points_transform are within this, x 1e-6, of its own in every entry (the nearest such frame)
 *   "tile_map"             how tiles are dealt to the eight XCDs: 2 (default) = 2x2 blocks of tiles round-robin, 1 = single
 *                          tiles round-robin (tile t on XCD t % 8), 0 = one contiguous band of tiles per XCD
 *   "speculative_binning"  1 (default) = R-dependent stages launched against a capacity, R read back late;
 *                          0 = read R, then launch
 *   "binning_capacity_hint" capacity of the NEXT forward, one shot (tests: force the overflow re-run).  The regular hints
 *                          are kept per caller -- (device, image size, half-octave bucket of N) -- so models of different
 *                          size alternating on one process (teacher / student, stage-A models) do not disturb each other
 *   "reset_speculation"    forget every capacity hint and zero the counters of gsr_get_counter
 *   "view_cache_reset"     (tests) every per-frame cache of the current device forgets its frames: the balanced placement's visit
 *                          counts, the list cut's remembered depths, the counters of gsr_debug_view_cache_stats /
 *                          gsr_debug_list_cut_stats (synchronises the device)
 *   "deterministic_backward" 1 = debug mode: the blend backward writes every (tile, Gaussian) partial gradient to its own slot and
 *                          a second kernel sums each Gaussian's slots in list order -- no float atomics, bit-identical gradients
 *                          from run to run (the default accumulates with atomics in arrival order); several times slower
 *                          (always through the two-pixel kernel: the four waves of "blend_bwd_ppt" 1 add into a tile's LDS row in arrival order)
 *   "profile"              1 = HIP events around every stage on the caller's stream (an event pair costs ~10 us of stream
 *                          bubble per stage), 2 = only the forward blend kernel is timed, through the start / stop timestamps of
 *                          its own dispatch (hipExtLaunchKernelGGL: still ~11 us of idle queue around the launch), 3 = as 2 on
 *                          every THIRD forward (bench.py's timed region: all of eight rotating views get sampled)
 *   "emit_hist"            1 (default) = the emission kernel counts the tile sort's digits itself (speculative flow); 0 = a
 *                          histogram kernel in front of the sort's passes.  Same lists either way (A/B and tests)
 *   "prep_hist_max_n"      "prepare in backward": up to this many Gaussians (default 262 144) the per-Gaussian backward kernel
 *                          also counts the next depth sort's digits.  Read by BOTH the backward that fills a hand-over buffer
 *                          and the forward that consumes it: do not change it between the two
 *   "poll_iters"           bound of gsr_forward's busy-wait on the pinned instance-count word, in units of ~50 ns (default
 *                          400 000 = 20 ms; past it the call waits with hipStreamSynchronize, which also reports a faulted
 *                          device); 0 = no busy waiting: an event is recorded behind the scan kernel and waited on (up to
 *                          round 3 that event was recorded on every forward: ~6 us of idle queue each) */
int gsr_set_option(const char* name, int value);
/* Monotonic counters: "spec_forwards" (forwards launched against a capacity), "spec_overflows" (of those, how many had to
 * re-run the binning because R exceeded the capacity), "exact_forwards" (read-then-launch forwards), "spec_callers"
 * (distinct (device, size, N bucket) entries); host-side time accounting: "forward_calls" / "forward_ns" (wall time inside
 * gsr_forward) / "forward_wait_ns" (the part of it spent waiting for the instance count) and "backward_calls" / "backward_ns" --
 * (forward_ns - forward_wait_ns + backward_ns) / calls is what the launching thread works per forward + backward.  -1 for an
 * unknown name. */
int64_t gsr_get_counter(const char* name);   /* + "blend_bwd_resident": workgroups of the backward blend the device holds at once; "depth_window_resorts" */
/* Debug / test hook: copy the per-tile ranges (T x {begin, end} uint32) and the (tile, depth, id)-ordered Gaussian-id list
 * (num_rendered uint32) out of a forward's binning buffer into device buffers of the caller (either may be NULL). */
int gsr_debug_read_binning(const void* binning, int64_t binning_capacity, int64_t num_rendered, int32_t W, int32_t H,
                           uint32_t* ranges_out, uint32_t* list_out, void* stream);
/* Debug / test hook: the geometry the direct binning (option "direct_binning", default on: tile lists by per-chunk tile histograms,
 * column scans and a one-wave-per-chunk scatter instead of emit + tile sort + ranges) would use for N Gaussians on T tiles:
 * out[0..6] = { 1 if the direct route serves this frame (0: the sort route -- more than 4 096 tiles, or N out of range),
 * Gaussians per chunk S (a multiple of 64), chunks NC, groups G, chunks per group Cg, T rounded up to 64, scratch bytes }.  Host only. */
int gsr_debug_direct_binning_geometry(int32_t N, int32_t T, int64_t out[7]);
/* The balanced placement's per-view cost caches of the CURRENT device for frames of W x H (synchronises the device):
 * out = {lookups, hits, entries in use, caches}.  A hit = the render found the entry of its frame (by GsrForwardArgs::view_id,
 * or by pose within "view_pose_tol_e6") and placed its waves by that frame's previous visit counts. */
int gsr_debug_view_cache_stats(int32_t W, int32_t H, int64_t out[4]);
/* The list cut's counters ("list_cut") of the CURRENT device for frames of W x H (synchronises the device): out = {renders that ran
 * the cut machinery, renders whose repair pass found flagged tiles, tiles repaired, chunks of the depth order behind the frame's cut
 * (summed), deep tiles (summed)}. */
int gsr_debug_list_cut_stats(int32_t W, int32_t H, int64_t out[5]);
/* Sum of the recorded durations of stage `name` ("preprocess_fwd", "sort_depth", "scan", "emit", "sort_tile",
 * "ranges", "blend_fwd", "blend_bwd", "preprocess_bwd") since the last read; synchronises on the events.  With the direct binning
 * "scan" = k_chunk_counts + the two column scans, "emit" = k_chunk_scatter, "sort_tile" / "ranges" record nothing. */
int gsr_profile_read(const char* name, double* total_ms, int64_t* count);

/* ---- "next" row f-3: fused photometric loss of the train step -----------------------
 * loss = (1-lambda)*mean|x-y| + lambda*(1-mean SSIM(x,y)), x = clamp(render,0,1) when clamp01_render != 0.
 * Replaces the torch ops of /root/reference/trainer/losses.py:98-136 (Loss.forward) and :147-209 (11x11 Gaussian
 * window SSIM) applied to the clamped render of /root/reference/scene/gaussian_model_ht.py:883.
 * out3 = {loss, mean ssim, mean l1}; workspace (gsr_loss_workspace_bytes) is kept for the backward.
 * grad_loss: device pointer to the upstream scalar gradient, or NULL for 1. */
size_t gsr_loss_workspace_bytes(int32_t C, int32_t H, int32_t W);
int gsr_loss_forward(const float* render, const float* target, int32_t C, int32_t H, int32_t W, float lambda_dssim,
                     int32_t clamp01_render, void* workspace, float* out3, void* stream);
int gsr_loss_backward(const float* render, const float* target, int32_t C, int32_t H, int32_t W, float lambda_dssim,
                      int32_t clamp01_render, const void* workspace, const float* grad_loss, float* d_render,
                      void* stream);

/* The same over a stack of `images` independent images [images, C, H, W] (a batched render, GsrBatch): every image is normalised by
 * its own C H W.  out3 = {SUM of the images' losses, mean SSIM, mean L1}; the backward returns the gradient of that sum, i.e. each
 * image receives exactly the gradient of its own loss. */
size_t gsr_loss_workspace_bytes_batched(int32_t images, int32_t C, int32_t H, int32_t W);
int gsr_loss_forward_batched(const float* render, const float* target, int32_t images, int32_t C, int32_t H, int32_t W, float lambda_dssim,
                             int32_t clamp01_render, void* workspace, float* out3, void* stream);
int gsr_loss_backward_batched(const float* render, const float* target, int32_t images, int32_t C, int32_t H, int32_t W, float lambda_dssim,
                              int32_t clamp01_render, const void* workspace, const float* grad_loss, float* d_render, void* stream);
/* gsr_loss_forward_batched with everything `Loss.forward` returns (/root/reference/trainer/losses.py:128-136) computed by the same
 * finishing kernel: out6 = {loss, mean SSIM, mean L1, loss_rgb = (1 - lambda) mean L1, loss_dssim = 1 - mean SSIM, loss_depth = 0}
 * -- the caller slices the vector instead of launching a torch kernel per term.  loss_copy (may be NULL): a second place for out6[0]
 * (the binding's differentiable scalar lives in storage of its own). */
int gsr_loss_forward_terms(const float* render, const float* target, int32_t images, int32_t C, int32_t H, int32_t W, float lambda_dssim,
                           int32_t clamp01_render, void* workspace, float* out6, float* loss_copy, void* stream);

/* ---- "next" row f-2: multi-tensor Adam step in one launch ------------------------------------------------
 * Same update rule as torch.optim.Adam(l, lr=0.0, eps=1e-15) of /root/reference/scene/gaussian_model_ht.py:275-289
 * (no amsgrad, no weight decay); `step` is the 1-based step count used for the bias corrections. */
#define GSR_ADAM_MAX_TENSORS 8
typedef struct GsrAdamTensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    uint64_t n;  /* elements */
    float lr;
} GsrAdamTensor;
int gsr_adam_step(const GsrAdamTensor* tensors, int32_t count, float beta1, float beta2, float eps, int64_t step,
                  void* stream);

/* Pose step of stage A (extension f-4; compute_relative_pose, trainer/ht3dgs_trainer.py:308-333): the pose being fitted is
 * M = Exp(delta) * base with six tangent numbers delta = (tau[3], phi[3]) under Adam -- `LieGroupParameter.retr()` +
 * torch.optim.Adam in the reference.  step >= 1: chains d_points_transform12 (dL/dM, as written by gsr_backward) to dL/d(delta),
 * applies torch's Adam update (lr, beta1, beta2, eps; bias correction for `step`) to delta6 / exp_avg6 / exp_avg_sq6 in place
 * and writes the new M (3x4 row-major) to points_transform_out12 -- where the next gsr_forward reads its points_transform.
 * step = 0: only evaluates M for the current delta.  base12 = NULL: identity.  All pointers device; one one-thread kernel. */
int gsr_pose_step(float* delta6, float* exp_avg6, float* exp_avg_sq6, const float* d_points_transform12, const float* base12,
                  float* points_transform_out12, float lr, float beta1, float beta2, float eps, int64_t step, void* stream);

/* The chain of gsr_pose_step alone (round 6): d_delta6_out = dL/d(delta) of M = Exp(delta) * base given dL/dM
 * (d_points_transform12, as written by gsr_backward), nothing updated.  It is the backward of the autograd node that stands where the
 * unmodified trainer evaluates `P[k].retr()` (scene/gaussian_model_ht.py:135-148): `.grad` of the frame's six pose numbers is
 * left where the trainer's own optimizer object looks for it (trainer/ht3dgs_trainer.py:162-166).  base12 = NULL: identity. */
int gsr_pose_grad(const float* delta6, const float* d_points_transform12, const float* base12, float* d_delta6_out, void* stream);

/* The same step when the pose lives in the render's CAMERA (the reference's camera_optimizer stepping a frame's pose after each
 * render of it, trainer/ht3dgs_trainer.py:162-166): viewmatrix = M^T, projmatrix = viewmatrix * projection_T, campos = -R^T t for
 * the world-to-camera M = Exp(delta) * base.  Takes gsr_backward's d_viewmatrix / d_projmatrix / d_campos (any may be NULL),
 * folds them into dL/dM, chains to dL/d(delta), applies Adam and rewrites the three camera tensors IN PLACE for the next render
 * of the frame.  projection_T16 = the transposed projection of the camera's intrinsics (projmatrix = viewmatrix * projection_T).
 * step = 0: only evaluates the camera tensors for the current delta. */
int gsr_pose_step_camera(float* delta6, float* exp_avg6, float* exp_avg_sq6, const float* d_viewmatrix16, const float* d_projmatrix16,
                         const float* d_campos3, const float* projection_T16, const float* base12, float* viewmatrix16,
                         float* projmatrix16, float* campos3, float lr, float beta1, float beta2, float eps, int64_t step, void* stream);

/* The trainer's per-iteration bookkeeping between backward() and optimizer.step() (trainer/ht3dgs_trainer.py:137-148), one launch
 * each instead of the reference's ~25 small torch kernels (what gsr_autopatch routes the unmodified trainer's statements to):
 *   gsr_masked_max         dst[i] = max(dst[i], (float)src[i]) where mask[i]: `max_radii2D[vis] = torch.max(max_radii2D[vis], radii[vis])`
 *   gsr_densify_stats_add  HTGaussianModel.add_densification_stats (scene/gaussian_model_ht.py:718-721): accum[i] += |grad[i, :2]|,
 *                          denom[i] += 1 where mask[i]; viewspace_grad3 = the [N,3] gradient of the screen-space points
 *   gsr_psnr               utils/image_utils.py:16-18 `psnr(img1, img2)`: out[c] = 20 log10(1 / sqrt(mean_c (a - b)^2)) for C planes of P
 *                          pixels each; scratch = gsr_psnr_scratch_bytes(C) bytes (two launches: partial sums, finish)
 * mask: one byte per element (torch.bool). */
int gsr_masked_max(float* dst, const int32_t* src, const uint8_t* mask, int32_t n, void* stream);
int gsr_densify_stats_add(float* xyz_gradient_accum, float* denom, const float* viewspace_grad3, const uint8_t* mask, int32_t n, void* stream);
size_t gsr_psnr_scratch_bytes(int32_t C);
int gsr_psnr(const float* a, const float* b, int32_t C, int64_t P, float* out, void* scratch, void* stream);

/* ---- "next" row f-1: simple_knn._C.distCUDA2 ----------------------------------------------------------------
 * out[i] = mean of the squared distances from points[i] to its 3 nearest other points (exact), the semantics of the
 * reference's SciPy twin /root/reference/scene/gaussian_model_ht.py:31-36; called at :211-216. */
size_t gsr_knn_scratch_bytes(int32_t N);
int gsr_knn_mean_dist2(const float* points /*[N,3]*/, int32_t N, float* out /*[N]*/, void* scratch, size_t scratch_bytes,
                       void* stream);

/* Measurement hook (bench.py roofline.peak_measured): float4 streaming copy of `bytes` bytes (a multiple of 16, 16-byte aligned
 * pointers) with `blocks` workgroups of 256 threads.  variant 0 = plain loads / stores, 1 = nt bit, 2 = four loads in flight per
 * lane (nt), 3 = persistent workgroups (a few hundred), contiguous runs, eight loads in flight per lane (nt).  The best achieved read +
 * write rate is the practical HBM ceiling of the box for a streaming kernel. */
int gsr_stream_copy(const void* src, void* dst, size_t bytes, int variant, int blocks, void* stream);

/* Building blocks exported for the unit tests of tests/test_gpu_blocks.py (device pointers). */
int gsr_sort_pairs_u32(uint32_t* keys, uint32_t* vals, uint32_t* keys_alt, uint32_t* vals_alt, uint32_t n,
                       int begin_bit, int end_bit, void* scratch, size_t scratch_bytes, int* result_in_alt,
                       void* stream);
int gsr_sort_pairs_u16(uint16_t* keys, uint32_t* vals, uint16_t* keys_alt, uint32_t* vals_alt, uint32_t n,
                       int begin_bit, int end_bit, void* scratch, size_t scratch_bytes, int* result_in_alt,
                       void* stream);
size_t gsr_sort_scratch_bytes(uint32_t n);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H */
