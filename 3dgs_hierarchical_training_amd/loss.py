"""Fused photometric loss (1-l)*L1 + l*(1-SSIM) of the train step, HIP kernels behind the C ABI.

Host-side mirror of what the reference computes with torch ops: `Loss.forward`
(/root/reference/trainer/losses.py:98-136) with the 11x11 Gaussian-window SSIM (:147-209), applied to the
clamped render (`rendered_image.clamp(0, 1)`, /root/reference/scene/gaussian_model_ht.py:883).  The clamp is
fused: pass the rasterizer's raw colour output.  No CPU path.
"""
import ctypes as C

import torch

from . import _lib as L


class _FusedPhotometricLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, render, target, lambda_dssim, clamp):
        lib = L.load()
        if render.device.type != "cuda":
            raise RuntimeError("fused_photometric_loss: tensors must be on a ROCm/HIP device (no CPU fallback)")
        render = render.float().contiguous()
        target = target.to(render.device).float().contiguous()
        Cn, H, W = render.shape
        ws = torch.empty(lib.gsr_loss_workspace_bytes(Cn, H, W), dtype=torch.uint8, device=render.device)
        out = torch.empty(3, dtype=torch.float32, device=render.device)
        with torch.cuda.device(render.device):
            st = torch.cuda.current_stream(render.device).cuda_stream
            L.check(lib.gsr_loss_forward(render.data_ptr(), target.data_ptr(), Cn, H, W, float(lambda_dssim), int(bool(clamp)),
                                         ws.data_ptr(), out.data_ptr(), C.c_void_p(st)), "gsr_loss_forward")
        ctx.save_for_backward(render, target, ws)
        ctx.cfg = (float(lambda_dssim), int(bool(clamp)))
        return out[0]

    @staticmethod
    def backward(ctx, grad_loss):
        lib = L.load()
        render, target, ws = ctx.saved_tensors
        lam, clamp = ctx.cfg
        Cn, H, W = render.shape
        d = torch.empty_like(render)
        g = grad_loss.float().contiguous()
        with torch.cuda.device(render.device):
            st = torch.cuda.current_stream(render.device).cuda_stream
            L.check(lib.gsr_loss_backward(render.data_ptr(), target.data_ptr(), Cn, H, W, lam, clamp, ws.data_ptr(),
                                          g.data_ptr(), d.data_ptr(), C.c_void_p(st)), "gsr_loss_backward")
        return d, None, None, None


def fused_photometric_loss(render: torch.Tensor, target: torch.Tensor, lambda_dssim: float = 0.2,
                           clamp: bool = True) -> torch.Tensor:
    """render: raw rasterizer colour [3,H,W] (clamped to [0,1] inside when clamp=True); target [3,H,W]."""
    return _FusedPhotometricLoss.apply(render, target, lambda_dssim, clamp)
