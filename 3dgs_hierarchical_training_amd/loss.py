"""Fused photometric loss (1-l)*L1 + l*(1-SSIM) of the train step, HIP kernels behind the C ABI.

Host-side mirror of what the reference computes with torch ops: `Loss.forward`
(/root/reference/trainer/losses.py:98-136) with the 11x11 Gaussian-window SSIM (:147-209), applied to the
clamped render (`rendered_image.clamp(0, 1)`, /root/reference/scene/gaussian_model_ht.py:883).  The clamp is
fused: pass the rasterizer's raw colour output.  No CPU path.
"""
import torch

from . import _ext as E


class _FusedPhotometricLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, render, target, lambda_dssim, clamp):
        if render.device.type != "cuda":
            raise RuntimeError("fused_photometric_loss: tensors must be on a ROCm/HIP device (no CPU fallback)")
        ops = E.load()
        render = render.float().contiguous()
        target = target.to(render.device).float().contiguous()
        out, ws = ops.photometric_loss_forward(render, target, float(lambda_dssim), bool(clamp))
        ctx.save_for_backward(render, target, ws)
        ctx.cfg = (float(lambda_dssim), bool(clamp))
        return out[0]

    @staticmethod
    def backward(ctx, grad_loss):
        render, target, ws = ctx.saved_tensors
        lam, clamp = ctx.cfg
        return E.load().photometric_loss_backward(render, target, ws, grad_loss, lam, clamp), None, None, None


def fused_photometric_loss(render: torch.Tensor, target: torch.Tensor, lambda_dssim: float = 0.2,
                           clamp: bool = True) -> torch.Tensor:
    """render: raw rasterizer colour [3,H,W] (clamped to [0,1] inside when clamp=True); target [3,H,W].
    One dispatcher call; the autograd node lives in the extension (csrc/torch_ext.cpp PhotometricLossFn) -- the Python
    autograd.Function above states the same thing and serves the plain-FFI binding route."""
    if render.device.type != "cuda":
        raise RuntimeError("fused_photometric_loss: tensors must be on a ROCm/HIP device (no CPU fallback)")
    if E.use_ctypes():
        return _FusedPhotometricLoss.apply(render, target, lambda_dssim, clamp)
    return E.load().photometric_loss(render, target, float(lambda_dssim), bool(clamp))


class _FusedPhotometricTerms(torch.autograd.Function):
    """The same op with its three results exposed: (loss, mean SSIM, mean L1).  Only `loss` carries a gradient."""

    @staticmethod
    def forward(ctx, render, target, lambda_dssim, clamp):
        ops = E.load()
        render = render.float().contiguous()
        target = target.to(render.device).float().contiguous()
        out, ws = ops.photometric_loss_forward(render, target, float(lambda_dssim), bool(clamp))
        ctx.save_for_backward(render, target, ws)
        ctx.cfg = (float(lambda_dssim), bool(clamp))
        loss, ssim_v, l1_v = out[0], out[1], out[2]
        ctx.mark_non_differentiable(ssim_v, l1_v)
        return loss, ssim_v, l1_v

    @staticmethod
    def backward(ctx, grad_loss, _g_ssim, _g_l1):
        render, target, ws = ctx.saved_tensors
        lam, clamp = ctx.cfg
        return E.load().photometric_loss_backward(render, target, ws, grad_loss.contiguous(), lam, clamp), None, None, None


def fused_photometric_loss_terms(render: torch.Tensor, target: torch.Tensor, lambda_dssim: float = 0.2, clamp: bool = True):
    """(loss, mean SSIM, mean L1): what `Loss.forward` of /root/reference/trainer/losses.py:98-136 reports as `loss`,
    `1 - loss_dssim` and `loss_rgb / (1 - lambda)` -- one fused forward, one fused backward (gsr_autopatch.loss_forward)."""
    if render.device.type != "cuda":
        raise RuntimeError("fused_photometric_loss: tensors must be on a ROCm/HIP device (no CPU fallback)")
    return _FusedPhotometricTerms.apply(render, target, lambda_dssim, clamp)


def fused_photometric_loss_report(render: torch.Tensor, target: torch.Tensor, lambda_dssim: float = 0.2, clamp: bool = True):
    """(loss, terms): the differentiable loss and the six-float vector {loss, mean SSIM, mean L1, loss_rgb = (1 - lambda) mean L1,
    loss_dssim = 1 - mean SSIM, loss_depth = 0} that the same finishing kernel wrote -- the whole return dict of `Loss.forward`
    (/root/reference/trainer/losses.py:128-136) from one dispatcher call; autograd node in the extension (PhotometricTermsFn),
    no gradient materialised for the vector.  Extension binding only."""
    if render.device.type != "cuda":
        raise RuntimeError("fused_photometric_loss: tensors must be on a ROCm/HIP device (no CPU fallback)")
    return E.load().photometric_loss_terms(render, target, float(lambda_dssim), bool(clamp))
