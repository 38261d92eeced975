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

  * `scene.gaussian_model_ht.CF3DGS_Render.render` (round 4; patched when that module is imported, now or later) hands the model's
    RAW parameter tensors (`_xyz, _features_dc, _features_rest, _opacity, _scaling, _rotation`) to `rasterize_gaussians_raw`: the
    wrapper's torch exp / sigmoid / normalize / cat (and their backward kernels) run inside the preprocess kernels, `get_xyz`'s
    pose action `P[idx].retr().act(xyz)` becomes the in-kernel `points_transform`, and the gradients arrive on the raw tensors
    directly, where `optimizer.step()` picks them up.  With the FusedAdam this module hands out, the backward kernel also computes
    that step's Adam update into SHADOW buffers (the model untouched) and `optimizer.step()` adopts it by swapping storages: the
    trainer's order -- densify / prune / opacity reset BEFORE the step, `update_gaussians=False`, the update dropped when
    `densify_and_prune` replaces every parameter (ht3dgs_trainer.py:137-160) -- is untouched, the separate 1.65 kB-per-Gaussian
    optimizer pass is gone.  (`GSR_AUTOPATCH_DEFERRED=0`: gradients to .grad and a one-launch FusedAdam.step() instead;
    `GSR_AUTOPATCH_DEFERRED_MIN_N`: the same for models below that many Gaussians, default 0.)  The returned dict is the
    reference's (`image` clamped, `depth`, `alpha`, `viewspace_points` whose .grad the backward fills, `visibility_filter`,
    `radii`).  `override_color`, `compute_cov3D_python`, `convert_SHs_python`, CPU tensors or an unexpected parameter layout
    fall back to the ORIGINAL method.  `GSR_AUTOPATCH_RENDER=0` leaves the method alone; `GSR_AUTOPATCH_POSE=0` keeps pose
    renders (rotate_xyz / rotate_seq) on the original method.
  * the frame poses (round 6): a lietorch SE3 `LieGroupParameter` that `get_xyz` reads (`P[k]`, gaussian_model_ht.py:135-148,
    :346-386) reaches the rasterizer through ONE autograd node over its six tangent numbers (`torch.ops.gsr.pose_matrix`: the [3,4]
    `points_transform` from one one-wave kernel, `.grad` of the parameter from another), and the pose optimizers the reference
    builds over such parameters -- `camera_optimizer[k]` (:296-311, stepped after every render: ht3dgs_trainer.py:162-166) and
    stage A's `training_setup_fix_position(gaussian_rot=False)` (:321-333) -- are `FusedPoseAdam` objects: one launch per
    `step()`.  The reference's ~50 small lietorch / torch launches per iteration (exponential map, group product, `matrix()`, their
    backward, Adam on six numbers) become three.  `GSR_AUTOPATCH_POSE_FUSED=0` keeps lietorch's chain and the stock Adam.
  * `HTGaussianModel.add_densification_stats` (same module) accumulates the same two sums without the boolean-mask gathers
    (each of them a `nonzero` + host synchronisation): one masked-add launch over N (gsr_densify_stats_add).
  * the render's `visibility_filter` is a bool-tensor subclass (`LazyMask`) under which the trainer's
    `max_radii2D[vis] = torch.max(max_radii2D[vis], radii[vis])` is one launch (gsr_masked_max) instead of three `nonzero`s;
    `utils.image_utils.psnr` (the training PSNR of every iteration) is two launches (gsr_psnr).  `GSR_AUTOPATCH_LAZY_MASK=0` /
    `GSR_AUTOPATCH_PSNR=0` switch those off.

`apply()` / `remove()` switch the patches on and off (importing the module calls `apply()`); `GSR_AUTOPATCH=0` disables them.
The rasterizer itself needs no patch: `diff_gaussian_rasterization` IS this library's drop-in package.
"""
import importlib
import importlib.abc
import importlib.machinery
import math
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
RENDER_MODULES = ("scene.gaussian_model_ht",)
_patched_render_classes = []        # [(class, attribute, original function)]
RAW_NAMES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
_REQUIRE_CUDA = True                # (tests drive the dispatch logic with CPU stand-ins and a recording rasterizer)


def _pkg():
    return importlib.import_module("3dgs_hierarchical_training_amd.optim"), importlib.import_module("3dgs_hierarchical_training_amd.loss")


def _ops():
    return importlib.import_module("3dgs_hierarchical_training_amd._ext").load()


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


def _wants_pose(params, kw) -> bool:
    """The reference's pose optimizers: `torch.optim.Adam([{'params': [P[k]], 'lr': ..., 'name': 'R'}], lr=0.0, eps=1e-15)` -- one per
    frame (`camera_optimizer[k]`, /root/reference/scene/gaussian_model_ht.py:296-311) or the model's `optimizer` during stage A's pose
    fit (`training_setup_fix_position(gaussian_rot=False)`, :321-333): every parameter a lietorch SE3 `LieGroupParameter` on the GPU."""
    if os.environ.get("GSR_AUTOPATCH_POSE_FUSED", "1") == "0":
        return False
    try:
        groups = list(params)
    except TypeError:
        return False
    if not groups or not all(isinstance(g, dict) for g in groups) or kw.get("amsgrad") or kw.get("weight_decay") or kw.get("maximize"):
        return False
    P = importlib.import_module("3dgs_hierarchical_training_amd.pose_opt")
    for g in groups:
        ps = [g["params"]] if torch.is_tensor(g["params"]) else list(g["params"])
        if not ps or not all(P.is_lie_pose(q) and (q.is_cuda or not _REQUIRE_CUDA) for q in ps):
            return False
    return True


class _AdamMeta(type(_ORIG_ADAM)):
    """`isinstance(opt, torch.optim.Adam)` keeps answering True for everything the patched name constructs (stock Adam objects and
    FusedAdam), as it did before the patch."""

    def __instancecheck__(cls, inst):
        if cls is _AdamDispatch:
            if isinstance(inst, _ORIG_ADAM):
                return True
            optim, pose_opt = sys.modules.get("3dgs_hierarchical_training_amd.optim"), sys.modules.get("3dgs_hierarchical_training_amd.pose_opt")
            return (optim is not None and isinstance(inst, optim.FusedAdam)) or (pose_opt is not None and isinstance(inst, pose_opt.FusedPoseAdam))
        return super().__instancecheck__(inst)


class _AdamDispatch(_ORIG_ADAM, metaclass=_AdamMeta):
    """`torch.optim.Adam` while the patch is applied: constructs FusedAdam for the reference's six-group optimizer, FusedPoseAdam
    for its pose optimizers (groups of lietorch SE3 parameters, `_wants_pose`) and the stock Adam for everything else.  (Returning an object that is not an instance of this class from __new__ skips __init__.)
    A SUBCLASS defined while the patch is applied (`class My(torch.optim.Adam)`) constructs normally: the dispatch only fires for
    the patched name itself."""

    def __new__(cls, params=None, *args, **kw):
        if cls is not _AdamDispatch:
            return super().__new__(cls)
        params = list(params)
        if _wants_fused(params, kw) and not args:
            optim, _ = _pkg()
            groups = [dict(g, params=[g["params"]] if torch.is_tensor(g["params"]) else list(g["params"])) for g in params]
            return optim.FusedAdam(groups, lr=kw.get("lr", 1e-3), betas=kw.get("betas", (0.9, 0.999)), eps=kw.get("eps", 1e-8))
        if _wants_pose(params, kw) and not args:
            P = importlib.import_module("3dgs_hierarchical_training_amd.pose_opt")
            groups = [dict(g, params=[g["params"]] if torch.is_tensor(g["params"]) else list(g["params"])) for g in params]
            return P.FusedPoseAdam(groups, lr=kw.get("lr", 1e-3), betas=kw.get("betas", (0.9, 0.999)), eps=kw.get("eps", 1e-8))
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
    raw = getattr(rgb_pred, "_gsr_raw", None)      # the patched render's un-clamped colour output (render_fused below), valid while
    fused_clamp = raw is not None and raw[1] == rgb_pred._version and raw[0].shape == rgb_pred.shape     # nobody wrote into the clamped image
    src = raw[0] if fused_clamp else rgb_pred
    if _ext_binding() and os.environ.get("GSR_AUTOPATCH_LOSS_REPORT", "1") != "0":
        # one dispatcher call: the finishing kernel wrote every entry of the returned dict (loss_rgb, loss_dssim, a zero loss_depth);
        # the entries are views of its six-float result -- no torch kernel per term, no zero fills in the backward
        loss, terms = loss_mod.fused_photometric_loss_report(src, rgb_gt, lambda_dssim, clamp=fused_clamp)
        rgb_full_loss, dssim_loss, zero_depth = terms[3], terms[4], terms[5]
    else:
        loss, ssim_v, l1_v = loss_mod.fused_photometric_loss_terms(src, rgb_gt, lambda_dssim, clamp=fused_clamp)
        rgb_full_loss = (1.0 - lambda_dssim) * l1_v
        dssim_loss = 1.0 - ssim_v
        zero_depth = None
    if lambda_depth != 0.0 and depth_pred is not None and depth_gt is not None:
        depth_gt = depth_gt.to(rgb_pred.device)
        depth_pred[depth_pred < 0.02] = 0.02
        depth_pred[depth_pred > 20.0] = 20.0
        depth_loss = self.get_depth_loss(depth_pred.squeeze(), depth_gt.squeeze())
        loss = loss + lambda_depth * depth_loss
    else:
        depth_loss = zero_depth if zero_depth is not None else torch.zeros((), device=rgb_pred.device)
    return {'loss': loss, 'loss_rgb': rgb_full_loss, 'loss_dssim': dssim_loss, 'loss_depth': depth_loss}


def _ext_binding() -> bool:
    E = importlib.import_module("3dgs_hierarchical_training_amd._ext")
    return not E.use_ctypes()


# ---- the visibility mask of the patched render: boolean-mask statements without their host synchronisations ------------------------
# The trainer's per-iteration statistics statement (/root/reference/trainer/ht3dgs_trainer.py:143-144),
#     gaussians.max_radii2D[visibility_filter] = torch.max(gaussians.max_radii2D[visibility_filter], radii[visibility_filter])
# is three boolean-mask index operations, each a `nonzero` with a host synchronisation that drains the launch queue -- +0.4 ms per
# iteration at every model size, in stage A as well (the statistics run on every iteration below `densify_until_iter`).  The mask
# this module's render returns is a bool tensor SUBCLASS that recognises exactly that pattern through `__torch_function__`:
# `dense[mask]` becomes a lazy selection (nothing gathered), `torch.max(selection, selection)` the element-wise maximum of the dense
# tensors, `dense[mask] = selection` one `torch.where` written in place -- the same values, no `nonzero`.  Anything else done with the
# mask or a selection (arithmetic, .sum(), printing, indexing other shapes ...) falls back to the plain tensors, gathered for real.
# GSR_AUTOPATCH_LAZY_MASK=0: the render returns a plain bool tensor.
class _LazySelection:
    """`dense[mask]` that has not been gathered.  Tensor-like only as far as the statement above needs; everything else
    materialises."""

    def __init__(self, dense, mask, max_of=None):
        self._dense, self.mask, self.max_of = dense, mask, max_of      # max_of = (a, b): the element-wise maximum, not yet evaluated
        # eager `dense[mask]` would have copied the values NOW: remember the version counters of what stands for them and refuse to
        # be evaluated after an in-place write to any of it -- the trainer's one statement never does that
        self._versions = [(t, t._version) for t in ((dense,) if dense is not None else tuple(max_of)) + (mask,)]

    def _check(self):
        for t, v in self._versions:
            if t._version != v:
                raise RuntimeError("gsr_autopatch: a tensor indexed by the render's visibility_filter was modified in place before the "
                                   "selection was used -- the lazy selection no longer stands for what `dense[mask]` would have copied; "
                                   "set GSR_AUTOPATCH_LAZY_MASK=0 (plain bool mask, eager gathers) for code that keeps such selections")

    @property
    def dense(self):
        self._check()
        if self._dense is None:
            self._dense = torch.maximum(*self.max_of)
            self._versions = [(self.mask, self._versions[-1][1])]
        return self._dense

    @property
    def dense_shape(self):
        return self.max_of[0].shape if self._dense is None else self._dense.shape

    def materialise(self):
        return self.dense[self.mask.as_subclass(torch.Tensor)]

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is torch.max and len(args) == 2 and not kwargs and isinstance(args[0], _LazySelection) and isinstance(args[1], _LazySelection) \
                and args[0].mask is args[1].mask and args[0].dense_shape == args[1].dense_shape:
            return _LazySelection(None, args[0].mask, max_of=(args[0].dense, args[1].dense))      # (.dense checks the versions)
        unwrap = lambda a: a.materialise() if isinstance(a, _LazySelection) else a
        return func(*[unwrap(a) for a in args], **{k: unwrap(v) for k, v in kwargs.items()})

    def __getattr__(self, name):          # methods / properties of the gathered tensor (.shape, .sum(), .float(), ...)
        return getattr(self.materialise(), name)

    def __repr__(self):
        return repr(self.materialise())


def _lazy_dunder(name):
    def f(self, *a, **k):
        return getattr(self.materialise(), name)(*a, **k)
    return f


for _n in ("add", "radd", "sub", "rsub", "mul", "rmul", "truediv", "rtruediv", "neg", "abs", "lt", "le", "gt", "ge", "eq", "ne", "len",
           "getitem", "iter", "float", "int", "bool", "and", "or", "invert", "pow", "matmul"):
    setattr(_LazySelection, f"__{_n}__", _lazy_dunder(f"__{_n}__"))


class LazyMask(torch.Tensor):
    """The `visibility_filter` of the patched render: a bool tensor that turns `dense[mask]` / `dense[mask] = selection` on a
    1-D tensor of its own length into their synchronisation-free forms (see above) and is an ordinary bool tensor otherwise."""

    @staticmethod
    def __new__(cls, mask):
        return torch.Tensor._make_subclass(cls, mask, False)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is torch.Tensor.__getitem__ and len(args) == 2 and isinstance(args[1], LazyMask) and type(args[0]) is torch.Tensor \
                and args[0].dim() == 1 and args[0].shape == args[1].shape and not kwargs:
            return _LazySelection(args[0], args[1])
        if func is torch.Tensor.__setitem__ and len(args) == 3 and isinstance(args[1], LazyMask) and isinstance(args[2], _LazySelection) \
                and args[2].mask is args[1] and type(args[0]) is torch.Tensor and args[0].shape == args[1].shape \
                and args[2].dense_shape == args[0].shape and not kwargs:
            dst, m, sel = args[0], args[1].as_subclass(torch.Tensor), args[2]
            sel._check()
            with torch._C.DisableTorchFunctionSubclass():
                pair = getattr(sel, "max_of", None)         # torch.max(dst[mask], radii[mask]): one launch (gsr_masked_max)
                if pair is not None and pair[0] is dst and dst.is_cuda and dst.dtype == torch.float32 and dst.is_contiguous() \
                        and pair[1].dtype == torch.int32 and pair[1].is_contiguous() and m.is_contiguous() and not dst.requires_grad:
                    _ops().masked_max_(dst, pair[1], m)
                else:
                    dst.copy_(torch.where(m, sel.dense.to(dst.dtype), dst))
            return None
        unwrap = lambda a: a.materialise() if isinstance(a, _LazySelection) else a
        with torch._C.DisableTorchFunctionSubclass():
            return func(*[unwrap(a) for a in args], **{k: unwrap(v) for k, v in kwargs.items()})


# ---- CF3DGS_Render.render on the raw-parameter path ------------------------------------------------------------------------------
def _pose_matrix(g):
    """The transform `get_xyz` applies to the means (/root/reference/scene/gaussian_model_ht.py:135-148) as a matrix that keeps its
    autograd link to the pose parameter, or None.  A lietorch SE3 `LieGroupParameter` on the GPU (what `init_RT` / `init_RT_seq` /
    `update_RT_seq` build, :346-386) goes through ONE autograd node over its six tangent numbers (`pose_opt.pose_matrix`: a [3,4]
    tensor, one kernel forward, one backward that leaves `.grad` on the parameter): no `retr()` chain, no matrix exponential in
    torch, nothing N-sized.  Any other pose object -- or `rotate_xyz_inverse`, or GSR_AUTOPATCH_POSE_FUSED=0 -- keeps the original
    statement: lietorch's `matrix()` of `retr()` ([4,4]; `matrix()` is `act` on the basis vectors)."""
    inverse = False
    if getattr(g, "rotate_xyz", False):
        p = g.P[0]
    elif getattr(g, "rotate_xyz_inverse", False):
        p, inverse = g.P[0], True
    elif getattr(g, "rotate_seq", False):
        p = g.P[g.seq_idx]
    else:
        return None
    if not inverse and os.environ.get("GSR_AUTOPATCH_POSE_FUSED", "1") != "0":
        P = importlib.import_module("3dgs_hierarchical_training_amd.pose_opt")
        if P.is_lie_pose(p) and (p.is_cuda or not _REQUIRE_CUDA):
            return P.pose_matrix(p, _ops())
    T = p.retr().inv() if inverse else p.retr()
    return T.matrix().reshape(4, 4)


_ZERO_POINTS = {}


def _view_id(cam, g) -> int:
    """A non-zero number that is the same whenever the same FRAME is rendered: the camera's `uid` (the frame index, or index + 0.5
    for an interpolated frame: /root/reference/trainer/trainer.py:556, :573) together with the pose slot `get_xyz` applies under
    `rotate_seq` (gaussian_model_ht.py:145-146).  The reference trains with an identity camera and moves the points, and steps the
    pose after every render, so neither the camera's matrices nor the pose's bits identify the frame; the id does.  Speed only
    (the forward blend's balanced placement, include/gsr.h GsrForwardArgs::view_id).  0 = unknown (GSR_AUTOPATCH_VIEW_ID=0: off)."""
    if os.environ.get("GSR_AUTOPATCH_VIEW_ID", "1") == "0":
        return 0
    try:
        vid = 1 + int(round(2.0 * float(getattr(cam, "uid"))))
        if getattr(g, "rotate_seq", False):
            vid += 1000003 * (1 + int(g.seq_idx))
        return vid if vid != 0 else 1
    except Exception:
        return 0


def _raw_tensors(g):
    """The six raw parameter tensors when they have the layout the fused path takes, else None."""
    ts = [getattr(g, k, None) for k in RAW_NAMES]
    if any(not torch.is_tensor(t) for t in ts):
        return None
    x = ts[0]
    if (_REQUIRE_CUDA and not x.is_cuda) or x.dim() != 2 or x.shape[0] == 0:
        return None
    N = x.shape[0]
    if any(t.dtype != torch.float32 or t.device != x.device or not t.is_contiguous() or t.shape[0] != N for t in ts):
        return None
    dc, rest, op, sc, rot = ts[1:]
    if tuple(x.shape) != (N, 3) or tuple(dc.shape) != (N, 1, 3) or rest.dim() != 3 or rest.shape[1] < 1 or rest.shape[2] != 3 or \
            tuple(op.shape) != (N, 1) or tuple(sc.shape) != (N, 3) or tuple(rot.shape) != (N, 4):
        return None
    return ts


def render_fused(self, viewpoint_camera, scaling_modifier=1.0, invert_bg_color=False, override_color=None,
                 compute_cov3D_python=False, convert_SHs_python=False):
    """Drop-in body of `CF3DGS_Render.render` (/root/reference/scene/gaussian_model_ht.py:775-908): same arguments, same returned
    dict.  The model's raw tensors go to the kernels as they are (activations of :128-133,176-188 and the SH concat in-kernel,
    `get_xyz`'s pose action as `points_transform`); every configuration the fused path does not cover runs the original method."""
    orig = next((f for c, a, f in _patched_render_classes if a == "render" and isinstance(self, c)), None)
    g = getattr(self, "gaussians", None)
    ts = None
    if override_color is None and not compute_cov3D_python and not convert_SHs_python and g is not None:
        ts = _raw_tensors(g)
    posed = ts is not None and (getattr(g, "rotate_xyz", False) or getattr(g, "rotate_xyz_inverse", False) or getattr(g, "rotate_seq", False))
    if posed and os.environ.get("GSR_AUTOPATCH_POSE", "1") == "0":
        ts = None
    if ts is None:
        if orig is None:
            raise RuntimeError("gsr_autopatch.render_fused: this configuration needs the original CF3DGS_Render.render")
        return orig(self, viewpoint_camera, scaling_modifier, invert_bg_color, override_color, compute_cov3D_python, convert_SHs_python)
    R = importlib.import_module("3dgs_hierarchical_training_amd.rasterizer")
    xyz, f_dc, f_rest, opacity, scaling, rotation = ts
    dev = xyz.device
    # the tensor whose .grad receives the 2D positional gradient (:800-808: `zeros_like(get_xyz, requires_grad=True) + 0` with
    # retain_grad); its values are never read by the rasterizer
    # (one zero buffer per (N, device), a fresh LEAF over it per render -- detach() makes a new tensor object on the same storage, no
    #  fill launch and no allocation; every render's leaf has its own .grad, and a leaf that requires grad cannot be written in place)
    # every render's leaf aliases this storage, so an in-place write under no_grad would reach all later renders -- the
    # buffer's version counter (shared by every detach() of it) must still be the one it was created with, else a fresh one is made
    # (a write through `.data` carries a counter of its own and is NOT seen: the values of `viewspace_points` are never read by the
    # rasterizer, only its .grad is written); and only the two most recent sizes are kept (a model and its frozen teacher), not one N x 12 B buffer per densification size
    zkey = (int(xyz.shape[0]), dev)
    zbuf = _ZERO_POINTS.get(zkey)
    if zbuf is None or zbuf._version != 0:
        while len(_ZERO_POINTS) >= 2:
            _ZERO_POINTS.pop(next(iter(_ZERO_POINTS)))
        _ZERO_POINTS.pop(zkey, None)
        zbuf = _ZERO_POINTS[zkey] = torch.zeros((xyz.shape[0], 3), dtype=torch.float32, device=dev)
    screenspace_points = zbuf.detach().requires_grad_(True)
    bg = self.bg_color if not invert_bg_color else 1 - self.bg_color
    settings = R.GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg,
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=g.active_sh_degree, campos=viewpoint_camera.camera_center,
        prefiltered=False, debug=False)
    M = _pose_matrix(g) if posed else None
    # DEFERRED optimizer step: when the model's optimizer is the FusedAdam this module handed out, the backward kernel also computes
    # the Adam update -- into shadow buffers, the model untouched -- and the trainer's `optimizer.step()` adopts it by swapping
    # storages.  Everything the trainer may do in between keeps its meaning (surgery sees the un-updated state and drops the
    # update; a skipped step leaves the model alone; stale .grad = torch's accumulation on the plain route).  GSR_AUTOPATCH_DEFERRED=0: off.
    opt, deferred = getattr(g, "optimizer", None), False
    if opt is not None and hasattr(opt, "deferred_ready") and torch.is_grad_enabled() and os.environ.get("GSR_AUTOPATCH_DEFERRED", "1") != "0":
        opt.flush_pending_as_grads()
        # (GSR_AUTOPATCH_DEFERRED_MIN_N: models below that many Gaussians keep the separate step; default 0 -- measured with the
        #  route's own order bias removed, the deferred step is the faster one at stage A's 130 k Gaussians too: 0.375 against
        #  0.40-0.42 ms per step; 0.90 against 0.98 ms at 1 M)
        if xyz.shape[0] >= int(os.environ.get("GSR_AUTOPATCH_DEFERRED_MIN_N", "0")):
            deferred = opt.deferred_ready({"xyz": xyz, "f_dc": f_dc, "f_rest": f_rest, "opacity": opacity, "scaling": scaling, "rotation": rotation})
    if _ext_binding() and os.environ.get("GSR_AUTOPATCH_EXTRAS", "1") != "0":
        # the clamped image and the visibility bytes come out of the kernels that hold the values (the blend's epilogue, the preprocess):
        # no torch launch for `clamp(0, 1)` / `radii > 0`
        image_raw, radii, depth, alpha, image, vis8 = R.rasterize_gaussians_raw(
            xyz, screenspace_points, f_dc, f_rest, opacity, scaling, rotation, settings, points_transform=M, fused_adam=opt if deferred else None,
            fused_adam_deferred=deferred, view_id=_view_id(viewpoint_camera, g), extras=3)
        visible = vis8.view(torch.bool) if vis8.numel() == radii.numel() else radii > 0
    else:
        image_raw, radii, depth, alpha = R.rasterize_gaussians_raw(xyz, screenspace_points, f_dc, f_rest, opacity, scaling, rotation,
                                                                   settings, points_transform=M, fused_adam=opt if deferred else None,
                                                                   fused_adam_deferred=deferred, view_id=_view_id(viewpoint_camera, g))
        image = image_raw.clamp(0, 1)
        visible = radii > 0
    image._gsr_raw = (image_raw, image._version)       # lets the patched Loss.forward fuse this clamp into the loss kernels
    if os.environ.get("GSR_AUTOPATCH_LAZY_MASK", "1") != "0":
        visible = LazyMask(visible)            # (the trainer's boolean-mask statistics statement without its host synchronisations)
    return {"image": image, "depth": depth, "alpha": alpha, "viewspace_points": screenspace_points,
            "visibility_filter": visible, "radii": radii}


def add_densification_stats_fused(self, viewspace_point_tensor, update_filter):
    """Drop-in body of `HTGaussianModel.add_densification_stats` (/root/reference/scene/gaussian_model_ht.py:718-721): the same
    sums as masked adds over N instead of three boolean-mask gathers / scatters (each a `nonzero` and a host synchronisation)."""
    if isinstance(update_filter, LazyMask):
        update_filter = update_filter.as_subclass(torch.Tensor)
    g = viewspace_point_tensor.grad
    if g is None or update_filter.dtype != torch.bool or update_filter.dim() != 1 or update_filter.shape[0] != self.denom.shape[0]:
        orig = next((f for c, a, f in _patched_render_classes if a == "add_densification_stats" and isinstance(self, c)), None)
        if orig is None:
            raise RuntimeError("gsr_autopatch.add_densification_stats_fused: needs a [N] boolean filter and a populated .grad")
        return orig(self, viewspace_point_tensor, update_filter)
    acc, den = self.xyz_gradient_accum, self.denom
    if g.is_cuda and all(t.dtype == torch.float32 and t.is_contiguous() and not t.requires_grad for t in (acc, den, g)) and \
            update_filter.is_contiguous() and acc.numel() == den.numel() == update_filter.shape[0] and tuple(g.shape) == (den.shape[0], 3):
        _ops().densify_stats_add_(acc, den, g, update_filter)       # one launch (gsr_densify_stats_add)
        return
    f = update_filter.unsqueeze(1)
    self.xyz_gradient_accum += torch.where(f, torch.norm(g[:, :2], dim=-1, keepdim=True), torch.zeros((), device=g.device, dtype=g.dtype))
    self.denom += f.to(self.denom.dtype)


def _patch_render_module(mod):
    if os.environ.get("GSR_AUTOPATCH_RENDER", "1") == "0":
        return
    for cname, attr, fn in (("CF3DGS_Render", "render", render_fused),
                            ("HTGaussianModel", "add_densification_stats", add_densification_stats_fused)):
        cls = getattr(mod, cname, None)
        if cls is None or not hasattr(cls, attr) or any(c is cls and a == attr for c, a, _ in _patched_render_classes):
            continue
        _patched_render_classes.append((cls, attr, getattr(cls, attr)))
        setattr(cls, attr, fn)


# ---- utils.image_utils.psnr: the training PSNR the trainer evaluates on every iteration (ht3dgs_trainer.py:138) ---------------------
IMAGE_UTILS_MODULES = ("utils.image_utils",)
_patched_psnr = []                  # [(module, attribute, original function)]


def psnr_fused(img1, img2):
    """Drop-in body of `psnr` (/root/reference/utils/image_utils.py:16-18): per-channel 20 log10(1 / sqrt(mse)), [C, 1], in two
    launches (gsr_psnr) instead of the statement's nine.  Anything but two same-shaped float32 GPU images runs the original."""
    if torch.is_tensor(img1) and torch.is_tensor(img2) and img1.is_cuda and img2.is_cuda and img1.dtype == torch.float32 and \
            img2.dtype == torch.float32 and img1.dim() >= 2 and img1.shape == img2.shape and img1.numel() > 0 and \
            not (torch.is_grad_enabled() and (img1.requires_grad or img2.requires_grad)):
        return _ops().psnr(img1, img2)
    orig = next((f for m, a, f in _patched_psnr if a == "psnr"), None)
    if orig is None:
        mse = (((img1 - img2)) ** 2).view(img1.shape[0], -1).mean(1, keepdim=True)
        return 20 * torch.log10(1.0 / torch.sqrt(mse))
    return orig(img1, img2)


def _patch_image_utils_module(mod):
    if os.environ.get("GSR_AUTOPATCH_PSNR", "1") == "0":
        return
    orig = getattr(mod, "psnr", None)
    if orig is None or orig is psnr_fused or any(m is mod for m, _, _ in _patched_psnr):
        return
    _patched_psnr.append((mod, "psnr", orig))
    mod.psnr = psnr_fused
    # modules that did `from utils.image_utils import psnr` before this patch hold the original under their own name
    for other in list(sys.modules.values()):
        if other is not None and other is not mod and getattr(other, "__dict__", {}).get("psnr") is orig:
            _patched_psnr.append((other, "psnr", orig))
            other.psnr = psnr_fused


def _patch_loss_module(mod):
    cls = getattr(mod, "Loss", None)
    if cls is None or any(c is cls for c, _ in _patched_loss_classes):
        return
    _patched_loss_classes.append((cls, cls.forward))
    cls.forward = loss_forward


class _PostImportFinder(importlib.abc.MetaPathFinder):
    """Patches `trainer.losses` / `scene.gaussian_model_ht` right after they have been executed, whenever that import happens."""

    def find_spec(self, fullname, path, target=None):
        if (fullname not in LOSS_MODULES and fullname not in RENDER_MODULES and fullname not in IMAGE_UTILS_MODULES) or not _applied:
            return None
        patch = _patch_loss_module if fullname in LOSS_MODULES else _patch_render_module if fullname in RENDER_MODULES \
            else _patch_image_utils_module
        for finder in sys.meta_path:
            if finder is self or not hasattr(finder, "find_spec"):
                continue
            spec = finder.find_spec(fullname, path, target)
            if spec is not None and spec.loader is not None and hasattr(spec.loader, "exec_module"):
                loader = spec.loader
                orig_exec = loader.exec_module

                def exec_module(module, _orig=orig_exec, _patch=patch):
                    _orig(module)
                    if _applied:
                        _patch(module)
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
    for name in RENDER_MODULES:
        if name in sys.modules:
            _patch_render_module(sys.modules[name])
    for name in IMAGE_UTILS_MODULES:
        if name in sys.modules:
            _patch_image_utils_module(sys.modules[name])


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
    while _patched_render_classes:
        cls, attr, fn = _patched_render_classes.pop()
        setattr(cls, attr, fn)
    while _patched_psnr:
        mod, attr, fn = _patched_psnr.pop()
        setattr(mod, attr, fn)


if os.environ.get("GSR_AUTOPATCH", "1") != "0":
    apply()


if __name__ == "__main__":      # python -m gsr_autopatch script.py [args...]: run a script (e.g. the reference's run.py) under the patches
    import runpy
    if len(sys.argv) < 2:
        raise SystemExit("usage: python -m gsr_autopatch <script.py> [args...]")
    sys.argv = sys.argv[1:]
    sys.path.insert(0, os.path.dirname(os.path.abspath(sys.argv[0])))
    runpy.run_path(sys.argv[0], run_name="__main__")
