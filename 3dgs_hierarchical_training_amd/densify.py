"""Adaptive density control around the rasterizer: the consumer of `means2D.grad` and `radii`.

Mirrors, without importing the reference, what HTGaussianModel does with the rasterizer's by-products:
  * statistics          /root/reference/scene/gaussian_model_ht.py:718-721 (`add_densification_stats`: norm of the first
                        two components of the screen-space gradient, accumulated per visible Gaussian) and the
                        `max_radii2D` update of /root/reference/trainer/ht3dgs_trainer.py:141-145
  * clone / split       gaussian_model_ht.py:632-676 (threshold on the mean gradient; small ones are cloned in place,
                        large ones replaced by two samples of themselves at scale / 1.6)
  * prune               gaussian_model_ht.py:678-691 (opacity floor, screen-size and world-size ceilings)
  * schedule            ht3dgs_trainer.py:137-155 (every `densification_interval` iterations after `densify_from_iter`,
                        opacity reset every `opacity_reset_interval`)
These are a few torch ops on N-sized tensors once per 100 iterations -- not a kernel.  The optimizer-state surgery is
GaussianParams' (train_step.py), which works on torch.optim.Adam and FusedAdam alike.

Ordering on the iterations that touch the parameters (round 4).  The reference collects the statistics and
densifies BETWEEN backward() and optimizer.step() (ht3dgs_trainer.py:135-160), and its surgery replaces parameter tensors by
fresh `nn.Parameter`s whose .grad is None -- `densify_and_prune` every one of them (`densification_postfix` runs even when nothing
is selected, gaussian_model_ht.py:584-629), `reset_opacity` the opacity tensor (:468-474) -- so the `optimizer.step()` that
follows finds nothing to step for them: THAT ITERATION'S ADAM UPDATE IS DROPPED (all six groups on a densification iteration,
the opacity group on a reset iteration; their step counts do not advance).  An Adam step fused into the backward kernel would
already have been applied by then.  `touches_parameters_at(iteration)` therefore tells `train_step` -- the schedule is known
before the render -- to run such an iteration UNFUSED: gradients to .grad, statistics, surgery, then a real `optimizer.step()`
on whatever still carries a gradient, exactly the reference's sequence.  One iteration in a hundred.
"""
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn.functional as F


@dataclass
class DensifyConfig:
    """Defaults of /root/reference/arguments/__init__.py:133-147."""
    percent_dense: float = 0.01
    densification_interval: int = 100
    opacity_reset_interval: int = 3000
    densify_from_iter: int = 500
    densify_until_iter: int = 15_000
    densify_grad_threshold: float = 0.0002
    min_opacity: float = 0.005
    reset_until_iter: Optional[int] = None
    max_points: Optional[int] = None      # harness-only safety bound (the reference has none)


def _quat_to_rotmat(q: torch.Tensor) -> torch.Tensor:
    """(w,x,y,z), normalised here -> [n,3,3] (the convention of /root/reference/utils/general_utils.py:76-99)."""
    q = F.normalize(q, dim=1)
    w, x, y, z = q.unbind(1)
    return torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)), dim=1).reshape(-1, 3, 3)


class Densifier:
    """Per-model statistics + the clone / split / prune surgery, on a `GaussianParams`."""

    def __init__(self, params, scene_extent: float = 1.0, cfg: Optional[DensifyConfig] = None, seed: int = 0):
        self.params = params
        self.extent = float(scene_extent)
        self.cfg = cfg or DensifyConfig()
        self.gen = torch.Generator(device=params._xyz.device).manual_seed(seed)
        self.reset_stats()

    def reset_stats(self):
        n, dev = self.params.num_points, self.params._xyz.device
        self.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
        self.denom = torch.zeros((n, 1), device=dev)
        self.max_radii2D = torch.zeros((n,), device=dev)

    # ---- gaussian_model_ht.py:718-721 + ht3dgs_trainer.py:141-147 -------------------------------------------------
    @torch.no_grad()
    def add_stats(self, viewspace_points: torch.Tensor, visibility_filter: torch.Tensor, radii: torch.Tensor):
        g = viewspace_points.grad
        if g is None:
            return
        vis = visibility_filter
        self.max_radii2D = torch.where(vis, torch.maximum(self.max_radii2D, radii.to(self.max_radii2D.dtype)), self.max_radii2D)
        self.xyz_gradient_accum += (g[:, :2].norm(dim=-1, keepdim=True)) * vis.unsqueeze(1)
        self.denom += vis.unsqueeze(1).to(self.denom.dtype)

    def fused_stats(self, iteration: int):
        """The three statistics tensors for the backward kernel to accumulate into (rasterize_gaussians_raw densify_stats), or
        None when this iteration collects none (iteration >= densify_until_iter, ht3dgs_trainer.py:137)."""
        if iteration >= self.cfg.densify_until_iter:
            return None
        return (self.xyz_gradient_accum, self.denom, self.max_radii2D)

    def touches_parameters_at(self, iteration: int) -> bool:
        """True when `after_backward(iteration, ...)` will replace parameter tensors (densify / prune, opacity reset): the schedule of
        ht3dgs_trainer.py:148-155, known before the render.  train_step keeps the optimizer out of the backward on such iterations
        (module docstring)."""
        c = self.cfg
        if iteration >= c.densify_until_iter:
            return False
        if iteration > c.densify_from_iter and iteration % c.densification_interval == 0:
            return True
        reset_until = c.reset_until_iter if c.reset_until_iter is not None else c.densify_until_iter
        return iteration % c.opacity_reset_interval == 0 and iteration < reset_until

    def _keep_stats(self, keep: torch.Tensor):
        self.xyz_gradient_accum = self.xyz_gradient_accum[keep]
        self.denom = self.denom[keep]
        self.max_radii2D = self.max_radii2D[keep]

    def _append(self, new):
        self.params.densification_postfix(new)
        self.reset_stats()          # gaussian_model_ht.py:621-629: statistics restart after every append

    def _prune(self, mask: torch.Tensor):
        self.params.prune_points(mask)
        self._keep_stats(~mask)

    # ---- gaussian_model_ht.py:632-691 --------------------------------------------------------------------------------
    @torch.no_grad()
    def densify_and_prune(self, max_grad: float, min_opacity: float, max_screen_size: Optional[float]):
        p = self.params
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        big = p.get_scaling.max(dim=1).values > self.cfg.percent_dense * self.extent

        # clone: under-reconstructed, small
        sel = (grads.norm(dim=-1) >= max_grad) & ~big
        if self.cfg.max_points is not None and p.num_points + int(sel.sum()) > self.cfg.max_points:
            sel = torch.zeros_like(sel)
        if bool(sel.any()):
            self._append({name: getattr(p, attr).detach()[sel] for name, attr in p._GROUP_ATTR.items()})
        else:
            self.reset_stats()      # the reference's densification_postfix runs (and zeroes the statistics) even with nothing selected

        # split: over-reconstructed, large (gradients of the Gaussians that were just appended count as zero)
        n = p.num_points
        padded = torch.zeros(n, device=grads.device)
        padded[:grads.shape[0]] = grads.squeeze(1)
        sel = (padded >= max_grad) & (p.get_scaling.max(dim=1).values > self.cfg.percent_dense * self.extent)
        if self.cfg.max_points is not None and n + int(sel.sum()) > self.cfg.max_points:
            sel = torch.zeros_like(sel)
        k = int(sel.sum())
        if k:
            stds = p.get_scaling.detach()[sel].repeat(2, 1)
            samples = torch.randn(stds.shape, generator=self.gen, device=stds.device) * stds
            rots = _quat_to_rotmat(p._rotation.detach()[sel]).repeat(2, 1, 1)
            new = {"xyz": torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + p._xyz.detach()[sel].repeat(2, 1),
                   "scaling": torch.log(p.get_scaling.detach()[sel].repeat(2, 1) / (0.8 * 2)),
                   "rotation": p._rotation.detach()[sel].repeat(2, 1),
                   "f_dc": p._features_dc.detach()[sel].repeat(2, 1, 1),
                   "f_rest": p._features_rest.detach()[sel].repeat(2, 1, 1),
                   "opacity": p._opacity.detach()[sel].repeat(2, 1)}
            self._append(new)
            self._prune(torch.cat((sel, torch.zeros(2 * k, dtype=torch.bool, device=sel.device))))
        else:
            self.reset_stats()      # (gaussian_model_ht.py:621-629 via :676: unconditional, so max_radii2D is always zero at the prune below)

        mask = (p.get_opacity.detach() < min_opacity).squeeze(1)
        if max_screen_size:
            mask = mask | (self.max_radii2D > max_screen_size) | (p.get_scaling.detach().max(dim=1).values > 0.1 * self.extent)
        if bool(mask.any()):
            self._prune(mask)

    # ---- ht3dgs_trainer.py:137-155 -------------------------------------------------------------------------------------
    @torch.no_grad()
    def after_backward(self, iteration: int, pkg, stats_done: bool = False) -> bool:
        """Call between backward() and optimizer.step() (drop-in order).  Returns True when the model was resized.
        stats_done: the backward kernel already accumulated this iteration's statistics (`fused_stats`)."""
        c = self.cfg
        if iteration >= c.densify_until_iter:
            return False
        if not stats_done:
            self.add_stats(pkg["viewspace_points"], pkg["visibility_filter"], pkg["radii"])
        resized = False
        if iteration > c.densify_from_iter and iteration % c.densification_interval == 0:
            size_threshold = 20 if iteration > c.opacity_reset_interval else None
            self.densify_and_prune(c.densify_grad_threshold, c.min_opacity, size_threshold)
            # the reference's surgery leaves EVERY parameter a fresh nn.Parameter without .grad, selected or not (module docstring):
            # the optimizer.step() that follows must find nothing to apply
            self.params.optimizer.zero_grad(set_to_none=True)
            resized = True
        reset_until = c.reset_until_iter if c.reset_until_iter is not None else c.densify_until_iter
        if iteration % c.opacity_reset_interval == 0 and iteration < reset_until:
            self.params.reset_opacity()
        return resized
