"""The benchmark's counterpart of the reference's training iteration around the rasterizer.

Mirrors, with stock torch ops outside the hot path (SURVEY.md section 8c last row):
  * parameter store + activations   /root/reference/scene/gaussian_model_ht.py:49-65,128-133,176-188
  * render wrapper                  gaussian_model_ht.py:775-894 (CF3DGS_Render.render, in-kernel SH / cov3D)
  * loss (1-l)*L1 + l*(1-SSIM)      /root/reference/trainer/losses.py:98-136,147-209, lambda_dssim = 0.2
                                    (/root/reference/arguments/__init__.py:134)
  * Adam(eps=1e-15), per-group LRs  gaussian_model_ht.py:263-289, arguments/__init__.py:116-131
  * train_step order                /root/reference/trainer/ht3dgs_trainer.py:81-169
It does not import the reference.  The rasterizer is the MI355X-native one (rasterizer.py -> C ABI -> HIP).
"""
import math
from typing import Dict

import torch
import torch.nn.functional as F

from ._ext import use_ctypes as E_use_ctypes
from .loss import fused_photometric_loss
from .optim import FusedAdam
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians_raw


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def _gauss_window(window_size: int, sigma: float, channel: int, device, dtype):
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    g = (g / g.sum()).unsqueeze(1)
    w2d = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
    return w2d.expand(channel, 1, window_size, window_size).contiguous().to(device=device, dtype=dtype)


_WINDOWS = {}


def ssim(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11) -> torch.Tensor:
    """11x11 Gaussian-window SSIM, sigma 1.5, zero padding, mean over the map (losses.py:147-209)."""
    if img1.dim() == 3:
        img1, img2 = img1.unsqueeze(0), img2.unsqueeze(0)
    ch = img1.shape[1]
    key = (ch, window_size, img1.device, img1.dtype)
    if key not in _WINDOWS:
        _WINDOWS[key] = _gauss_window(window_size, 1.5, ch, img1.device, img1.dtype)
    w = _WINDOWS[key]
    pad = window_size // 2
    mu1 = F.conv2d(img1, w, padding=pad, groups=ch)
    mu2 = F.conv2d(img2, w, padding=pad, groups=ch)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = F.conv2d(img1 * img1, w, padding=pad, groups=ch) - mu1_sq
    s2 = F.conv2d(img2 * img2, w, padding=pad, groups=ch) - mu2_sq
    s12 = F.conv2d(img1 * img2, w, padding=pad, groups=ch) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return m.mean()


def photometric_loss(pred: torch.Tensor, gt: torch.Tensor, lambda_dssim: float = 0.2) -> torch.Tensor:
    l1 = torch.abs(pred - gt).mean()
    return (1.0 - lambda_dssim) * l1 + lambda_dssim * (1.0 - ssim(pred, gt))


class GaussianParams:
    """Raw (pre-activation) parameters, laid out as HTGaussianModel keeps them."""

    def __init__(self, scene: Dict, device, spatial_lr_scale: float = 1.0, optimizer: str = "hip"):
        d = device
        self.max_sh_degree = int(round(math.sqrt(scene["shs"].shape[1]))) - 1
        self.active_sh_degree = int(scene["sh_degree"])
        self._xyz = scene["means3D"].to(d).clone().requires_grad_(True)
        self._features_dc = scene["shs"][:, :1].to(d).clone().contiguous().requires_grad_(True)
        self._features_rest = scene["shs"][:, 1:].to(d).clone().contiguous().requires_grad_(True)
        self._scaling = torch.log(scene["scales"].to(d)).requires_grad_(True)
        self._rotation = scene["rotations"].to(d).clone().requires_grad_(True)
        self._opacity = inverse_sigmoid(scene["opacities"].to(d).clamp(1e-4, 1 - 1e-4)).requires_grad_(True)
        self._build_optimizer(spatial_lr_scale, optimizer)

    def _build_optimizer(self, spatial_lr_scale: float, optimizer: str):
        groups = [
            {"params": [self._xyz], "lr": 0.00016 * spatial_lr_scale, "name": "xyz"},
            {"params": [self._features_dc], "lr": 0.0025, "name": "f_dc"},
            {"params": [self._features_rest], "lr": 0.0025 / 20.0, "name": "f_rest"},
            {"params": [self._opacity], "lr": 0.05, "name": "opacity"},
            {"params": [self._scaling], "lr": 0.005, "name": "scaling"},
            {"params": [self._rotation], "lr": 0.001, "name": "rotation"},
        ]
        # same update rule as the reference's torch.optim.Adam(l, lr=0.0, eps=1e-15); `fused` only selects the
        # single-pass implementation (one kernel per group instead of the foreach chain)
        if optimizer == "hip":        # gsr_adam_step: all six groups in one HIP launch
            self.optimizer = FusedAdam(groups, lr=0.0, eps=1e-15)
        elif optimizer == "torch_fused":
            self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15, fused=self._xyz.is_cuda)
        else:                         # the reference's own construction (foreach implementation)
            self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)

    @classmethod
    def from_raw(cls, seg: Dict[str, torch.Tensor], device, sh_degree: int = 3, spatial_lr_scale: float = 1.0,
                 optimizer: str = "hip") -> "GaussianParams":
        """Model from the six raw tensors (`_xyz`, `_features_dc`, ... as HTGaussianModel.capture() holds them,
        gaussian_model_ht.py:92-104) with a fresh optimizer -- what `training_setup` does after a merge
        (/root/reference/trainer/ht3dgs_trainer.py:793)."""
        self = cls.__new__(cls)
        self.max_sh_degree = int(round(math.sqrt(seg["_features_rest"].shape[1] + 1))) - 1
        self.active_sh_degree = int(sh_degree)
        for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
            setattr(self, k, seg[k].detach().to(device=device, dtype=torch.float32).contiguous().clone().requires_grad_(True))
        self._build_optimizer(spatial_lr_scale, optimizer)
        return self

    def raw(self) -> Dict[str, torch.Tensor]:
        """The six raw tensors, detached (no copy)."""
        return {k: getattr(self, k).detach() for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")}

    @property
    def num_points(self):
        return self._xyz.shape[0]

    def oneup_sh_degree(self) -> int:
        """`oneupSHdegree` (/root/reference/scene/gaussian_model_ht.py:193-195): the reference starts every model at active degree 0
        (:68) with all (max_sh_degree + 1)^2 coefficients stored and calls this once per 1 000 global iterations
        (/root/reference/trainer/ht3dgs_trainer.py:580-581); renders use the ACTIVE degree (:818)."""
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1
        return self.active_sh_degree

    # ---- optimizer-state surgery of densification / pruning / opacity reset -----------------------------------------
    # Same protocol as HTGaussianModel (/root/reference/scene/gaussian_model_ht.py:532-629): the parameter tensor of a
    # group is replaced by a new leaf and its Adam moments are sliced / zero-extended / zeroed with it.  Works on
    # torch.optim.Adam and on FusedAdam alike (both key `state` by the parameter tensor).
    _GROUP_ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
                   "scaling": "_scaling", "rotation": "_rotation"}

    def _swap_group_tensor(self, group, new_tensor, moments):
        """moments: callable(old_state_tensor) -> new state tensor (shape of new_tensor)."""
        old = group["params"][0]
        st = self.optimizer.state.pop(old, None)
        new = new_tensor.detach().contiguous().requires_grad_(True)
        if st is not None:
            st["exp_avg"], st["exp_avg_sq"] = moments(st["exp_avg"]), moments(st["exp_avg_sq"])
            self.optimizer.state[new] = st
        group["params"][0] = new
        setattr(self, self._GROUP_ATTR[group["name"]], new)
        self._prepared = None            # a hand-over buffer of "prepare in backward" describes the old tensors
        return new

    def prune_points(self, mask: torch.Tensor):
        """Remove the Gaussians where mask is True (gaussian_model_ht.py:568-582)."""
        keep = ~mask
        for g in self.optimizer.param_groups:
            self._swap_group_tensor(g, g["params"][0][keep], lambda m: m[keep].contiguous())
        self._screenspace_zero = None

    def densification_postfix(self, new: Dict[str, torch.Tensor]):
        """Append Gaussians; `new` maps group name -> tensor of new rows (gaussian_model_ht.py:584-629)."""
        for g in self.optimizer.param_groups:
            ext = new[g["name"]]
            self._swap_group_tensor(g, torch.cat((g["params"][0].detach(), ext), dim=0),
                                    lambda m, ext=ext: torch.cat((m, torch.zeros_like(ext)), dim=0))
        self._screenspace_zero = None

    def reset_opacity(self, ceiling: float = 0.01):
        """opacity <- min(opacity, ceiling) with zeroed moments (gaussian_model_ht.py:468-474, 532-546)."""
        for g in self.optimizer.param_groups:
            if g["name"] == "opacity":
                new = inverse_sigmoid(torch.min(self.get_opacity.detach(), torch.full_like(self._opacity, ceiling)))
                self._swap_group_tensor(g, new, torch.zeros_like)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return F.normalize(self._rotation)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)


def make_settings(scene: Dict, device, sh_degree: int, bg=None) -> GaussianRasterizationSettings:
    bg = scene["bg"] if bg is None else bg
    return GaussianRasterizationSettings(
        image_height=int(scene["image_height"]), image_width=int(scene["image_width"]),
        tanfovx=float(scene["tanfovx"]), tanfovy=float(scene["tanfovy"]), bg=bg.to(device).float(),
        scale_modifier=1.0, viewmatrix=scene["viewmatrix"].to(device), projmatrix=scene["projmatrix"].to(device),
        sh_degree=sh_degree, campos=scene["campos"].to(device), prefiltered=False, debug=False)


class _LazyVisibility(dict):
    """The render package of CF3DGS_Render.render (gaussian_model_ht.py:886-894); `visibility_filter = radii > 0` is a kernel
    launch nobody reads on most iterations, so it is evaluated on first access."""

    def __missing__(self, key):
        if key == "visibility_filter":
            v = self["visibility_filter"] = self["radii"] > 0
            return v
        raise KeyError(key)

    # the other ways a dict is read see the key as well (a drop-in consumer may use any of them)
    def get(self, key, default=None):
        if key == "visibility_filter":
            return self[key]
        return super().get(key, default)

    def __contains__(self, key):
        return key == "visibility_filter" or super().__contains__(key)

    def _materialise(self):
        self["visibility_filter"]
        return self

    def keys(self):
        return dict.keys(self._materialise())

    def items(self):
        return dict.items(self._materialise())

    def values(self):
        return dict.values(self._materialise())

    def __iter__(self):
        return dict.__iter__(self._materialise())

    def __len__(self):
        return dict.__len__(self._materialise())


def with_sh_degree(settings: GaussianRasterizationSettings, degree: int) -> GaussianRasterizationSettings:
    """The same view at another active SH degree (CF3DGS_Render.render builds its settings from `active_sh_degree` every call,
    gaussian_model_ht.py:818); the tensors are shared, so `_same_view` recognises the result as the same camera."""
    return settings if int(settings.sh_degree) == int(degree) else settings._replace(sh_degree=int(degree))


def _same_view(a: GaussianRasterizationSettings, b: GaussianRasterizationSettings) -> bool:
    if a is b:
        return True
    return all((x is y) if torch.is_tensor(x) else (x == y) for x, y in zip(a, b))


class PoseState:
    """One camera pose under refinement, the way the reference keeps it (`self.P[k]`, a lietorch SE3 parameter with its own Adam,
    gaussian_model_ht.py:296-311, stepped after every render of its frame, ht3dgs_trainer.py:162-166): the pose is the transform
    M = Exp(delta) * base applied to the Gaussians' means in-kernel (`points_transform`), delta = six tangent numbers.  `step()`
    is ONE kernel (gsr_pose_step): dL/dM -> dL/d(delta), Adam on delta, next M written in place."""

    def __init__(self, base_w2c: torch.Tensor, device, lr: float = 1e-3, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-15):
        from . import _ext
        self._ops = _ext.load()
        self.base = base_w2c.detach().float()[:3].contiguous().to(device)
        self.delta = torch.zeros(6, device=device)
        self.m, self.v = torch.zeros(6, device=device), torch.zeros(6, device=device)
        self.M = torch.zeros(3, 4, device=device)
        self.lr, self.b1, self.b2, self.eps, self.steps = lr, beta1, beta2, eps, 0
        self._none = torch.empty(0, device=device)
        self._ops.pose_step(self.delta, self.m, self.v, self._none, self.base, self.M, lr, beta1, beta2, eps, 0)
        self._leaf = None
        self.frozen = False          # True: rendered through its transform like the others, never stepped (the gauge frame)

    def leaf(self) -> torch.Tensor:
        """The [3,4] transform as a fresh autograd leaf for one render (its .grad is what `step` consumes)."""
        self._leaf = self.M.detach().requires_grad_(not self.frozen)
        return self._leaf

    def step(self):
        if self.frozen or self._leaf is None or self._leaf.grad is None:
            self._leaf = None
            return
        self.steps += 1
        self._ops.pose_step(self.delta, self.m, self.v, self._leaf.grad, self.base, self.M, self.lr, self.b1, self.b2, self.eps, self.steps)
        self._leaf = None

    def matrix(self) -> torch.Tensor:
        """Current world-to-camera [4,4] on the host."""
        out = torch.eye(4)
        out[:3] = self.M.detach().cpu()
        return out


class CameraPoseState:
    """A frame's pose under refinement, kept in the CAMERA of its renders: `settings` is the frame's raster-settings tuple whose
    viewmatrix / projmatrix / campos are this object's own device tensors (leaves that receive d_viewmatrix / d_projmatrix /
    d_campos from the backward), functions of M = Exp(delta) * base (world-to-camera).  `step()` -- ONE kernel,
    gsr_pose_step_camera -- folds the three gradients into dL/dM, chains to the six tangent numbers, applies Adam and rewrites
    the three tensors in place: the reference's `camera_optimizer[fidx].step()` after a render of frame fidx
    (ht3dgs_trainer.py:162-166; Adam eps 1e-15, gaussian_model_ht.py:296-311).  View-dependent colour keeps world-frame
    directions (PoseState, the transform-of-the-means route, evaluates it in each camera's own frame, as the reference's
    get_xyz route does)."""

    def __init__(self, template: GaussianRasterizationSettings, base_w2c: torch.Tensor, device, lr: float = 1e-3, beta1: float = 0.9,
                 beta2: float = 0.999, eps: float = 1e-15):
        from . import _ext
        self._ops = _ext.load()
        vm0, pm0 = template.viewmatrix.detach().double().cpu(), template.projmatrix.detach().double().cpu()
        self.projT = (torch.linalg.inv(vm0) @ pm0).float().contiguous().to(device)      # projmatrix = viewmatrix @ projection_T
        self.base = base_w2c.detach().float()[:3].contiguous().to(device)
        self.delta = torch.zeros(6, device=device)
        self.m, self.v = torch.zeros(6, device=device), torch.zeros(6, device=device)
        self.vm = torch.zeros(4, 4, device=device, requires_grad=True)
        self.pm = torch.zeros(4, 4, device=device, requires_grad=True)
        self.cp = torch.zeros(3, device=device, requires_grad=True)
        self.lr, self.b1, self.b2, self.eps, self.steps = lr, beta1, beta2, eps, 0
        self.frozen = False
        self._none = torch.empty(0, device=device)
        with torch.no_grad():
            self._ops.pose_step_camera(self.delta, self.m, self.v, self._none, self._none, self._none, self.projT, self.base,
                                       self.vm, self.pm, self.cp, lr, beta1, beta2, eps, 0)
        self.settings = template._replace(viewmatrix=self.vm, projmatrix=self.pm, campos=self.cp)

    def freeze(self):
        """The gauge frame: rendered like the others, never stepped, no camera gradients asked for."""
        self.frozen = True
        for t in (self.vm, self.pm, self.cp):
            t.requires_grad_(False)

    def step(self):
        g = (self.vm.grad, self.pm.grad, self.cp.grad)
        if self.frozen or all(x is None for x in g):
            self.vm.grad = self.pm.grad = self.cp.grad = None
            return
        self.steps += 1
        with torch.no_grad():
            self._ops.pose_step_camera(self.delta, self.m, self.v, *(self._none if x is None else x for x in g), self.projT, self.base,
                                       self.vm, self.pm, self.cp, self.lr, self.b1, self.b2, self.eps, self.steps)
        self.vm.grad = self.pm.grad = self.cp.grad = None

    def matrix(self) -> torch.Tensor:
        """Current world-to-camera [4,4] on the host."""
        return self.vm.detach().t().contiguous().cpu()


def render(params: GaussianParams, settings: GaussianRasterizationSettings, clamp: bool = True,
           fused_activations: bool = False, fused_adam=None, next_settings: GaussianRasterizationSettings = None,
           points_transform: torch.Tensor = None, next_points_transform: torch.Tensor = None, densify_stats=None,
           view_id: int = 0) -> Dict:
    """CF3DGS_Render.render with compute_cov3D_python = convert_SHs_python = False.
    view_id: the frame's id (non-zero; the reference's `viewpoint_camera.uid`) -- speed only (rasterize_gaussians_raw).
    fused_activations=True hands the raw parameters to the kernels (exp / sigmoid / normalize / cat in-kernel).
    next_settings (with fused_adam): the camera of the NEXT render of this model -- its preprocess then rides in this render's
    backward ("prepare in backward", rasterize_gaussians_raw) and the hand-over buffer is kept on `params` until a render with
    that camera picks it up (single use; any parameter surgery drops it)."""
    settings = with_sh_degree(settings, params.active_sh_degree)     # gaussian_model_ht.py:818: sh_degree = the model's ACTIVE degree
    xyz = params.get_xyz
    # gaussian_model_ht.py:800-805 builds `zeros_like(xyz, requires_grad=True) + 0` every render only to receive
    # the 2D positional gradient; its VALUES are never read by the rasterizer.  One zero leaf per model does the same
    # job without a fill + add over N x 3 floats per step: its .grad is reset here and filled by backward().
    screenspace_points = getattr(params, "_screenspace_zero", None)
    if screenspace_points is None or screenspace_points.shape != xyz.shape or screenspace_points.device != xyz.device:
        screenspace_points = torch.zeros_like(xyz, requires_grad=True)
        params._screenspace_zero = screenspace_points
    screenspace_points.grad = None
    bfb = getattr(params, "first_block", None)       # batched.BatchedGaussianParams: B models, B cameras, B images
    if bfb is not None and not fused_activations:
        raise RuntimeError("a batch of models renders through the raw-parameter path (fused_activations=True)")
    if fused_activations:
        prep, use = getattr(params, "_prepared", None), None
        if prep is not None:
            params._prepared = None          # single use: the forward sorts the buffer's keys in place
            if prep["valid"] and prep["n"] == xyz.shape[0] and prep["xyz"] is params._xyz and _same_view(prep["settings"], settings) \
                    and prep.get("cam") == _camera_versions(settings) and _same_transform(prep.get("xf"), points_transform):
                use = prep["buf"]
        want_next = next_settings if (fused_adam is not None and next_settings is not None) else None
        out = rasterize_gaussians_raw(xyz, screenspace_points, params._features_dc, params._features_rest, params._opacity,
                                      params._scaling, params._rotation, settings, fused_adam=fused_adam, prepared=use,
                                      prepare_next=want_next, points_transform=points_transform,
                                      next_points_transform=next_points_transform if want_next is not None else None,
                                      densify_stats=densify_stats, batch_first_block=bfb, view_id=view_id)
        if want_next is not None:        # filled by this render's backward; train_step marks it valid once that has run
            nxf = next_points_transform if next_points_transform is not None else points_transform
            params._prepared = {"buf": out[4], "settings": want_next, "n": xyz.shape[0], "xyz": params._xyz, "valid": False,
                                "xf": None if nxf is None else (nxf.data_ptr(), nxf._version), "cam": _camera_versions(want_next)}
            out = out[:4]
    else:
        rasterizer = GaussianRasterizer(raster_settings=settings)
        out = rasterizer(means3D=xyz, means2D=screenspace_points, shs=params.get_features, colors_precomp=None,
                         opacities=params.get_opacity, scales=params.get_scaling, rotations=params.get_rotation,
                         cov3D_precomp=None)
    rendered_image, radii, rendered_depth, rendered_alpha = out
    return _LazyVisibility({"image": rendered_image.clamp(0, 1) if clamp else None, "raw_image": rendered_image,
                            "depth": rendered_depth, "alpha": rendered_alpha, "viewspace_points": screenspace_points, "radii": radii})


def _camera_versions(rs) -> tuple:
    """Version counters of a settings tuple's camera tensors: a pose step that rewrites them in place (CameraPoseState.step) makes
    a hand-over buffer prepared for the old values stale."""
    return (rs.viewmatrix._version, rs.projmatrix._version, rs.campos._version)


def _same_transform(tag, xf) -> bool:
    """The hand-over buffer was prepared with this very pose transform: same storage, not written since (a PoseState.step() of
    that frame in between bumps the version)."""
    if tag is None or xf is None:
        return tag is None and xf is None
    return tag == (xf.data_ptr(), xf._version)


def train_step(params: GaussianParams, settings: GaussianRasterizationSettings, gt: torch.Tensor,
               lambda_dssim: float = 0.2, fused_loss: bool = True, fused_activations: bool = True,
               fused_optimizer: bool = True, densifier=None, iteration: int = 0, next_settings=None,
               pose: "PoseState" = None, next_pose: "PoseState" = None, next_sh_degree: int = None, view_id: int = 0) -> Dict:
    """render -> loss -> backward -> Adam step (ht3dgs_trainer.py:102-166 without densification).
    fused_loss=True evaluates clamp + L1 + SSIM in the HIP loss kernels; False uses the torch restatement.
    fused_activations=True runs exp / sigmoid / normalize / cat inside the rasterizer kernels.
    fused_optimizer=True (needs fused_activations and the "hip" optimizer) applies the Adam step inside the
    per-Gaussian backward kernel -- same update, the gradients just never travel through HBM; optimizer.step()
    then finds no .grad and is a no-op.
    next_settings: the camera the NEXT train_step of this model will use, when the caller knows it (a trainer draws its frame
    one step ahead): the backward then also runs the next render's preprocess on the updated parameters and the next forward
    skips that kernel (same result bit for bit; see rasterize_gaussians_raw).
    The render uses the model's ACTIVE SH degree (`params.active_sh_degree`, as CF3DGS_Render.render does); next_sh_degree = the
    degree of the next step when the caller is about to raise it (`oneup_sh_degree()` right after this step: the reference does so
    when global_iteration % 1000 == 0) -- the hand-over then already carries the colours of the higher degree.
    densifier (densify.Densifier) + iteration: the adaptive density control of ht3dgs_trainer.py:137-155 runs between
    backward() and optimizer.step(), as in the reference; on the iterations where it replaces parameter tensors the step is not
    fused into the backward, so that the update the reference drops there is dropped here too (densify.py)."""
    fused_adam = params.optimizer if (fused_optimizer and fused_activations and isinstance(params.optimizer, FusedAdam)) else None
    # an iteration whose `after_backward` replaces parameter tensors (densify / prune, opacity reset) runs UNFUSED: the reference's
    # surgery sits between backward() and optimizer.step() and drops that iteration's update of the tensors it replaces
    # (ht3dgs_trainer.py:137-160; densify.py) -- an update applied inside the backward kernel could not be dropped any more
    if fused_adam is not None and densifier is not None and densifier.touches_parameters_at(iteration):
        fused_adam = None
    # pose refinement (the reference's camera_optimizer, ht3dgs_trainer.py:162-166): this frame's transform is an autograd leaf of
    # the render; the hand-over to the next render is skipped when that render is of THIS frame (its transform changes in between)
    cam_pose = isinstance(pose, CameraPoseState)
    if cam_pose:             # the pose is this frame's camera: its settings are the render's settings
        settings = pose.settings
        if next_pose is not None:
            next_settings = next_pose.settings
    xf = pose.leaf() if (pose is not None and not cam_pose) else None
    deg = int(params.active_sh_degree)
    settings = with_sh_degree(settings, deg)
    # the hand-over needs the 16-coefficient layout (max_sh_degree 3, the reference's); any active degree, and the next render may
    # be one degree up
    nxt = next_settings if params.max_sh_degree == 3 else None
    if nxt is not None:
        ndeg = deg if next_sh_degree is None else int(next_sh_degree)
        nxt = with_sh_degree(nxt, ndeg) if ndeg in (deg, deg + 1) and ndeg <= 3 else None
    if pose is not None and nxt is not None and (next_pose is None or (next_pose is pose and not pose.frozen)):
        nxt = None               # the next render is of THIS frame, whose pose moves in between: no hand-over
    # per-iteration densification statistics (max_radii2D, xyz_gradient_accum, denom: ht3dgs_trainer.py:141-147) accumulate inside
    # the per-Gaussian backward kernel when the densifier offers its tensors (raw-parameter path through the extension)
    dstats = densifier.fused_stats(iteration) if (densifier is not None and fused_activations and not E_use_ctypes()) else None
    pkg = render(params, settings, clamp=not fused_loss, fused_activations=fused_activations, fused_adam=fused_adam,
                 next_settings=nxt, points_transform=xf, densify_stats=dstats, view_id=view_id,
                 next_points_transform=next_pose.M if (xf is not None and next_pose is not None and nxt is not None) else None)
    if fused_loss:          # (a batch hands [B,3,H,W] stacks: the fused loss is then the SUM of the models' losses, every image
        loss = fused_photometric_loss(pkg["raw_image"], gt, lambda_dssim, clamp=True)     # normalised on its own -- each model gets
    else:                   # exactly its own loss's gradient, bit-identical with training it alone)
        loss = photometric_loss(pkg["image"], gt, lambda_dssim)
    # same as loss.backward(); the upstream "1" is kept on the device instead of being filled by a launch every step
    one = getattr(params, "_grad_one", None)
    if one is None or one.device != loss.device:
        one = params._grad_one = torch.ones((), dtype=loss.dtype, device=loss.device)
    loss.backward(gradient=one)
    prep = getattr(params, "_prepared", None)
    if prep is not None and not prep["valid"]:
        prep["valid"] = True             # the backward that fills the hand-over buffer has been enqueued
    if pose is not None:
        pose.step()
    if densifier is not None:
        densifier.after_backward(iteration, pkg, stats_done=dstats is not None)
    params.optimizer.step()
    params.optimizer.zero_grad(set_to_none=True)
    pkg["loss"] = loss.detach()
    return pkg
