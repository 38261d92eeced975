"""Loader of the PyTorch-ROCm extension (csrc/torch_ext.cpp -> csrc/torch_build/gsr_torch.so) and the fake kernels that
make its ops traceable (torch.compile / FakeTensor).

`torch.ops.gsr.*` is the product's binding: GaussianRasterizer, the fused loss, FusedAdam and simple_knn go through
it.  The ctypes view of the same C ABI (_lib.py) stays as the plain-FFI example of INTEGRATION.md and for the debug /
profiling hooks; `GSR_BINDING=ctypes` routes the rasterizer through it (one GPU test does, to keep it honest).
No fallback: a missing gsr_torch.so raises.
"""
import os

import torch

from . import _lib as L

HERE = os.path.dirname(os.path.abspath(__file__))
EXT_PATH = os.path.join(HERE, "csrc", "torch_build", "gsr_torch.so")
_loaded = False


def use_ctypes() -> bool:
    return os.environ.get("GSR_BINDING", "") == "ctypes"


def load():
    """torch.ops.load_library(gsr_torch.so) once; returns torch.ops.gsr."""
    global _loaded
    if not _loaded:
        L.load()        # libgsr_hip.so first (fails loudly when it is not built)
        if not os.path.exists(EXT_PATH):
            raise RuntimeError(f"{EXT_PATH} not found: the PyTorch extension is not built (run `python __graft_entry__.py` or "
                               "`python 3dgs_hierarchical_training_amd/build.py`).  There is no CPU fallback.")
        torch.ops.load_library(EXT_PATH)
        _register_fakes()
        _loaded = True
    return torch.ops.gsr


def _register_fakes():
    lib = L.load()

    @torch.library.register_fake("gsr::rasterize_forward")
    def _(means3D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, sh_rest, viewmatrix, projmatrix, campos, bg,
          points_transform, image_height, image_width, tanfovx, tanfovy, scale_modifier, sh_degree, raw_params, prefiltered, debug,
          prepared, batch_first_block, view_id=0, extras=0):
        N, H, W = means3D.shape[0], image_height, image_width
        f = lambda *s: means3D.new_empty(s, dtype=torch.float32)
        b = lambda n: means3D.new_empty((n,), dtype=torch.uint8)
        ctx = torch.library.get_ctx()
        nbin = ctx.new_dynamic_size()          # the binning buffer is R-sized: data dependent
        B = len(batch_first_block) - 1 if len(batch_first_block) >= 3 else 1
        lead = (B,) if B > 1 else ()
        return (f(*lead, 3, H, W), means3D.new_empty((N,), dtype=torch.int32), f(*lead, 1, H, W), f(*lead, 1, H, W), b(lib.gsr_geom_bytes(int(N))),
                b(lib.gsr_image_bytes_batched(int(W), int(H), B)), b(nbin), torch.empty((3,), dtype=torch.int64),
                f(*lead, 3, H, W) if (extras & 1) else f(0), b(N if ((extras & 2) and prepared.numel() == 0) else 0))

    @torch.library.register_fake("gsr::rasterize")
    def _(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, sh_rest, viewmatrix, projmatrix, campos,
          bg, points_transform, image_height, image_width, tanfovx, tanfovy, scale_modifier, sh_degree, raw_params, prefiltered, debug,
          cam_grad, adam_m, adam_v, adam_lr, beta1, beta2, eps, step, prepared, next_viewmatrix, next_projmatrix, next_campos,
          next_height, next_width, next_tanfovx, next_tanfovy, next_points_transform, next_sh_degree, adam_commit, densify_stats, batch_first_block, view_id=0, extras=0):
        N, H, W = means3D.shape[0], image_height, image_width
        f = lambda *s: means3D.new_empty(s, dtype=torch.float32)
        nprep = lib.gsr_prepared_bytes(int(N)) if next_viewmatrix.numel() else 0
        B = len(batch_first_block) - 1 if len(batch_first_block) >= 3 else 1
        lead = (B,) if B > 1 else ()
        return (f(*lead, 3, H, W), means3D.new_empty((N,), dtype=torch.int32), f(*lead, 1, H, W), f(*lead, 1, H, W),
                means3D.new_empty((nprep,), dtype=torch.uint8), f(*lead, 3, H, W) if (extras & 1) else f(0),
                means3D.new_empty((N if ((extras & 2) and prepared.numel() == 0) else 0,), dtype=torch.uint8))

    @torch.library.register_fake("gsr::rasterize_backward")
    def _(means3D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, sh_rest, viewmatrix, projmatrix, campos, bg,
          points_transform, geom, image, binning, meta, grad_color, grad_depth, grad_alpha, image_height, image_width, tanfovx, tanfovy,
          scale_modifier, sh_degree, raw_params, need_viewmatrix, need_projmatrix, need_campos, need_points_transform, densify_stats, radii,
          batch_first_block):
        N = means3D.shape[0]
        f = lambda *s: means3D.new_empty(s, dtype=torch.float32)
        has = lambda t: t.numel() > 0
        M = (sh.shape[1] + (sh_rest.shape[1] if has(sh_rest) else 0)) if has(sh) else 0
        none = f(0)
        B = len(batch_first_block) - 1 if len(batch_first_block) >= 3 else 1
        lead = (B,) if B > 1 else ()           # a batch of B models: one camera gradient per model (as the real op returns them)
        return [f(N, 3), f(N, 3), f(N, 1 if has(sh_rest) else M, 3) if has(sh) else none, f(N, 3) if has(colors_precomp) else none,
                f(N, 1), f(N, 3) if has(scales) else none, f(N, 4) if has(scales) else none, f(N, 6) if has(cov3D_precomp) else none,
                f(N, M - 1, 3) if (has(sh) and has(sh_rest)) else none, f(*lead, 4, 4) if need_viewmatrix else none,
                f(*lead, 4, 4) if need_projmatrix else none, f(*lead, 3) if need_campos else none,
                f(*lead, 3, 4) if (need_points_transform and has(points_transform)) else none]

    @torch.library.register_fake("gsr::rasterize_backward_fused")
    def _(means3D, sh, sh_rest, opacities, scales, rotations, viewmatrix, projmatrix, campos, bg, points_transform, geom, image, binning,
          meta, grad_color, grad_depth, grad_alpha, image_height, image_width, tanfovx, tanfovy, scale_modifier, sh_degree,
          need_viewmatrix, need_projmatrix, need_campos, need_points_transform, adam_m, adam_v, adam_lr, beta1, beta2, eps, step,
          next_viewmatrix, next_projmatrix, next_campos, next_height, next_width, next_tanfovx, next_tanfovy, prepared_out,
          next_points_transform, next_sh_degree, densify_stats, radii, batch_first_block):
        f = lambda *s: means3D.new_empty(s, dtype=torch.float32)
        none = f(0)
        B = len(batch_first_block) - 1 if len(batch_first_block) >= 3 else 1
        lead = (B,) if B > 1 else ()
        return [f(means3D.shape[0], 3), f(*lead, 4, 4) if need_viewmatrix else none, f(*lead, 4, 4) if need_projmatrix else none,
                f(*lead, 3) if need_campos else none, f(*lead, 3, 4) if (need_points_transform and points_transform.numel() > 0) else none]

    # in-place ops without a return value: nothing to describe beyond the schema's (a!) annotations
    @torch.library.register_fake("gsr::adam_step")
    def _(params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step):
        return None

    @torch.library.register_fake("gsr::pose_step")
    def _(delta, exp_avg, exp_avg_sq, d_xf, base, xf, lr, beta1, beta2, eps, step):
        return None

    @torch.library.register_fake("gsr::pose_step_camera")
    def _(delta, exp_avg, exp_avg_sq, d_viewmatrix, d_projmatrix, d_campos, projection_T, base, viewmatrix, projmatrix, campos, lr,
          beta1, beta2, eps, step):
        return None

    @torch.library.register_fake("gsr::pose_matrix_forward")
    def _(delta, base):
        return delta.new_empty((3, 4), dtype=torch.float32)

    @torch.library.register_fake("gsr::pose_matrix")
    def _(delta, base):
        return delta.new_empty((3, 4), dtype=torch.float32)

    @torch.library.register_fake("gsr::pose_matrix_backward")
    def _(delta, base, d_xf):
        return torch.empty_like(delta)

    @torch.library.register_fake("gsr::masked_max_")
    def _(dst, src, mask):
        return None

    @torch.library.register_fake("gsr::densify_stats_add_")
    def _(accum, denom, grad, mask):
        return None

    @torch.library.register_fake("gsr::psnr")
    def _(a, b):
        return a.new_empty((a.shape[0], 1), dtype=torch.float32)

    @torch.library.register_fake("gsr::mark_visible")
    def _(means3D, viewmatrix, projmatrix):
        return means3D.new_empty((means3D.shape[0],), dtype=torch.bool)

    @torch.library.register_fake("gsr::photometric_loss_forward")
    def _(render, target, lambda_dssim, clamp):
        C, H, W = render.shape[-3:]
        B = render.shape[0] if render.dim() == 4 else 1
        return render.new_empty((3,), dtype=torch.float32), render.new_empty((lib.gsr_loss_workspace_bytes(int(B * C), int(H), int(W)),), dtype=torch.uint8)

    @torch.library.register_fake("gsr::photometric_loss")
    def _(render, target, lambda_dssim, clamp):
        return render.new_empty((), dtype=torch.float32)

    @torch.library.register_fake("gsr::photometric_loss_terms")
    def _(render, target, lambda_dssim, clamp):
        return render.new_empty((), dtype=torch.float32), render.new_empty((6,), dtype=torch.float32)

    @torch.library.register_fake("gsr::photometric_loss_backward")
    def _(render, target, workspace, grad_loss, lambda_dssim, clamp):
        return torch.empty_like(render, dtype=torch.float32)

    @torch.library.register_fake("gsr::knn_mean_dist2")
    def _(points):
        return points.new_empty((points.shape[0],), dtype=torch.float32)
