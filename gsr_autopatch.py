"""gsr_autopatch -- opt-in fast path for the UNMODIFIED reference trainer.

The reference builds its optimizer and its loss from stock torch pieces:
  * `torch.optim.Adam(l, lr=0.0, eps=1e-15)` over six groups named xyz / f_dc / f_rest / opacity / scaling / rotation
    (/root/reference/scene/gaussian_model_ht.py:275-289) -> foreach Adam: ~1 ms per step at 1 M Gaussians;
  * `Loss.forward` = (1 - l) L1 + l (1 - SSIM) with an 11x11 depthwise `F.conv2d` SSIM
    (/root/reference/trainer/losses.py:98-136, :164-251) -> ~2.7 ms per step at 980x545.
Importing this module BEFORE the trainer swaps both for the HIP kernels of this library without touching a reference file:

    python -m gsr_autopatch run.py --mode train --config arguments/full/Tanks/Francis.yml      # runs run.py under the patches
    # or, at the top of your own launcher:  import gsr_autopatch

  * `torch.optim.Adam(...)` whose parameter groups carry exactly those six names returns a `FusedAdam` (same `param_groups` /
    `state` / `state_dict` protocol, so the model's prune / densify / opacity-reset surgery and capture / restore run on it
    unchanged; one HIP launch per `step()`); every other Adam construction is the stock class;
  * `trainer.losses.Loss.forward` (patched when that module is imported, now or later) evaluates L1 + SSIM forward and backward
    in the fused loss kernels (gsr_loss_forward / gsr_loss_backward) and returns the same dict (`loss`, `loss_rgb`,
    `loss_dssim`, `loss_depth`); the depth term, when enabled, stays the reference's own code.

`apply()` / `remove()` switch the patches on and off (importing the module calls `apply()`); `GSR_AUTOPATCH=0` disables them.
The rasterizer itself needs no patch: `diff_gaussian_rasterization` IS this library's drop-in package.
"""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

FUSED_NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
_ORIG_ADAM = torch.optim.Adam
_applied = False
_patched_loss_classes = []          # [(class, original forward)]
LOSS_MODULES = ("trainer.losses",)


def _pkg():
    return importlib.import_module("3dgs_hierarchical_training_amd.optim"), importlib.import_module("3dgs_hierarchical_training_amd.loss")


def _wants_fused(params, kw) -> bool:
    """The reference's construction: a list of group dicts, one tensor each, named exactly the six groups; no amsgrad, no weight
    decay, parameters on the GPU."""
    try:
        groups = list(params)
    except TypeError:
        return False
    if len(groups) != 6 or not all(isinstance(g, dict) for g in groups):
        return False
    if sorted(g.get("name", "") for g in groups) != sorted(FUSED_NAMES):
        return False
    if kw.get("amsgrad") or kw.get("weight_decay") or kw.get("maximize"):
        return False
    for g in groups:
        ps = list(g["params"]) if not torch.is_tensor(g["params"]) else [g["params"]]
        if len(ps) != 1 or not ps[0].is_cuda or ps[0].dtype != torch.float32:
            return False
    return True


class _AdamDispatch(_ORIG_ADAM):
    """`torch.optim.Adam` while the patch is applied: constructs FusedAdam for the reference's six-group optimizer and the stock
    Adam for everything else.  (Returning an object that is not an instance of this class from __new__ skips __init__.)"""

    def __new__(cls, params, *args, **kw):
        params = list(params)
        if _wants_fused(params, kw) and not args:
            optim, _ = _pkg()
            groups = [dict(g, params=[g["params"]] if torch.is_tensor(g["params"]) else list(g["params"])) for g in params]
            return optim.FusedAdam(groups, lr=kw.get("lr", 1e-3), betas=kw.get("betas", (0.9, 0.999)), eps=kw.get("eps", 1e-8))
        return _ORIG_ADAM(params, *args, **kw)


def loss_forward(self, rgb_pred, rgb_gt, depth_pred=None, depth_gt=None, rgb_loss_type='l1', **kwargs):
    """Drop-in body of `trainer.losses.Loss.forward` (/root/reference/trainer/losses.py:98-136): same arguments, same returned
    dict; the photometric part is ONE fused forward (+ one fused backward) instead of ~25 torch kernels."""
    _, loss_mod = _pkg()
    lambda_dssim = float(self.cfg.lambda_dssim)
    lambda_depth = float(getattr(self.cfg, "lambda_depth", 0.0))
    rgb_gt = rgb_gt.to(rgb_pred.device)
    if rgb_pred.dim() != 3 or not rgb_pred.is_cuda:       # not the [3,H,W] device image the fused kernels take: the original code
        orig = next((f for c, f in _patched_loss_classes if isinstance(self, c)), None)
        if orig is None:
            raise RuntimeError("gsr_autopatch.loss_forward: needs a [C,H,W] image on the GPU")
        return orig(self, rgb_pred, rgb_gt, depth_pred, depth_gt, rgb_loss_type, **kwargs)
    loss, ssim_v, l1_v = loss_mod.fused_photometric_loss_terms(rgb_pred, rgb_gt, lambda_dssim, clamp=False)
    rgb_full_loss = (1.0 - lambda_dssim) * l1_v
    dssim_loss = 1.0 - ssim_v
    if lambda_depth != 0.0 and depth_pred is not None and depth_gt is not None:
        depth_gt = depth_gt.to(rgb_pred.device)
        depth_pred[depth_pred < 0.02] = 0.02
        depth_pred[depth_pred > 20.0] = 20.0
        depth_loss = self.get_depth_loss(depth_pred.squeeze(), depth_gt.squeeze())
        loss = loss + lambda_depth * depth_loss
    else:
        depth_loss = torch.zeros((), device=rgb_pred.device)
    return {'loss': loss, 'loss_rgb': rgb_full_loss, 'loss_dssim': dssim_loss, 'loss_depth': depth_loss}


def _patch_loss_module(mod):
    cls = getattr(mod, "Loss", None)
    if cls is None or any(c is cls for c, _ in _patched_loss_classes):
        return
    _patched_loss_classes.append((cls, cls.forward))
    cls.forward = loss_forward


class _PostImportFinder(importlib.abc.MetaPathFinder):
    """Patches `trainer.losses` right after it has been executed, whenever that import happens."""

    def find_spec(self, fullname, path, target=None):
        if fullname not in LOSS_MODULES or not _applied:
            return None
        for finder in sys.meta_path:
            if finder is self or not hasattr(finder, "find_spec"):
                continue
            spec = finder.find_spec(fullname, path, target)
            if spec is not None and spec.loader is not None and hasattr(spec.loader, "exec_module"):
                loader = spec.loader
                orig_exec = loader.exec_module

                def exec_module(module, _orig=orig_exec):
                    _orig(module)
                    if _applied:
                        _patch_loss_module(module)
                try:
                    loader.exec_module = exec_module
                except Exception:
                    return None
                return spec
        return None


_FINDER = _PostImportFinder()


def apply():
    """Switch the patches on (idempotent)."""
    global _applied
    if _applied:
        return
    _applied = True
    torch.optim.Adam = _AdamDispatch       # (the reference looks the class up as `torch.optim.Adam` at call time, gaussian_model_ht.py:289)
    if _FINDER not in sys.meta_path:
        sys.meta_path.insert(0, _FINDER)
    for name in LOSS_MODULES:
        if name in sys.modules:
            _patch_loss_module(sys.modules[name])


def remove():
    """Switch the patches off again (tests, A/B runs)."""
    global _applied
    if not _applied:
        return
    _applied = False
    torch.optim.Adam = _ORIG_ADAM
    if _FINDER in sys.meta_path:
        sys.meta_path.remove(_FINDER)
    while _patched_loss_classes:
        cls, fwd = _patched_loss_classes.pop()
        cls.forward = fwd


if os.environ.get("GSR_AUTOPATCH", "1") != "0":
    apply()


if __name__ == "__main__":      # python -m gsr_autopatch script.py [args...]: run a script (e.g. the reference's run.py) under the patches
    import runpy
    if len(sys.argv) < 2:
        raise SystemExit("usage: python -m gsr_autopatch <script.py> [args...]")
    sys.argv = sys.argv[1:]
    sys.path.insert(0, os.path.dirname(os.path.abspath(sys.argv[0])))
    runpy.run_path(sys.argv[0], run_name="__main__")
