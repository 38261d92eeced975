"""Stand-ins shaped like the reference's live objects, for the tests and the bench leg of `gsr_autopatch.render_fused`.

The reference's `CF3DGS_Render` / `HTGaussianModel` / `Camera` (/root/reference/scene/gaussian_model_ht.py:47-90,726-773,
/root/reference/scene/cameras.py:17-98) import lietorch, plyfile and the dataset stack, none of which exists in this image or on the
GPU box; these classes carry exactly the ATTRIBUTES the render wrapper and the train step read from them -- no behaviour of their own
beyond the reference's one-line activations -- so that the patched method can be driven the way the unmodified trainer drives it
(/root/reference/trainer/ht3dgs_trainer.py:102-166).  tests/helpers/ref_render_driver.py runs the same method on the real classes
in the authoring container.
"""
import math

import torch


class StubCamera:
    """FoVx / FoVy / image size / the three matrices, as `Camera` keeps them (cameras.py:60-98); built from a scene dict of
    synthetic.make_scene or from raw matrices."""

    def __init__(self, W, H, tanfovx, tanfovy, viewmatrix, projmatrix, campos, uid=0, original_image=None):
        self.image_width, self.image_height = int(W), int(H)
        self.FoVx, self.FoVy = 2.0 * math.atan(float(tanfovx)), 2.0 * math.atan(float(tanfovy))
        self.world_view_transform, self.full_proj_transform, self.camera_center = viewmatrix, projmatrix, campos
        self.uid = uid
        self.original_image = original_image

    @classmethod
    def from_scene(cls, scene, device, original_image=None, uid=0):
        return cls(scene["image_width"], scene["image_height"], scene["tanfovx"], scene["tanfovy"], scene["viewmatrix"].to(device),
                   scene["projmatrix"].to(device), scene["campos"].to(device), uid, original_image)


class SE3:
    """Stand-in for `lietorch.SE3` as far as the reference's model file uses it (/root/reference/scene/gaussian_model_ht.py:135-166,
    346-386): `SE3(pose7)` with `.data` = [..., 7] (tx ty tz qx qy qz qw), `tangent_shape`, `retr(a)` = Exp(a) * self,
    `act(points)`, `matrix()`, `inv()`, `*`.  lietorch is a third-party CUDA extension that exists neither in this image nor on the
    GPU box; this class states its PUBLIC behaviour with plain torch (pose.py's closed forms), so that the trainer's pose
    statements can be driven -- and timed as a torch chain -- without it.  An element that came out of an operation carries its
    matrix only (`.data` is then None): enough for `act` / `matrix` / `inv` / `*`, which is all the reference asks of one."""

    def __init__(self, data=None, _matrix=None):
        self.data = data
        self._matrix = _matrix

    @property
    def tangent_shape(self):
        return tuple(self.data.shape[:-1]) + (6,)

    def _M(self):
        if self._matrix is None:
            from . import pose
            return pose.pose7_to_matrix(self.data.reshape(7))
        return self._matrix

    def retr(self, a):
        from . import pose
        return SE3(_matrix=pose.se3_exp(a.reshape(6)) @ self._M())

    def matrix(self):
        return self._M()[None]

    def inv(self):
        M = self._M()
        Rt = M[:3, :3].t()
        top = torch.cat((Rt, -(Rt @ M[:3, 3:4])), dim=1)
        return SE3(_matrix=torch.cat((top, M[3:4]), dim=0))

    def act(self, x):
        M = self._M()
        return x @ M[:3, :3].t() + M[:3, 3]

    def __mul__(self, other):
        return SE3(_matrix=self._M() @ other._M())


class LieGroupParameter(torch.Tensor):
    """Stand-in for `lietorch.LieGroupParameter`: a float32 tensor SUBCLASS of the group's tangent shape, zeros at construction,
    `__torch_function__` disabled, the group element in `.group`; `retr()` = Exp(self) * group (lietorch/groups.py, public API).
    A stock torch.optim.Adam updates the six tangent numbers in place (addcdiv_), the group element stays."""
    __torch_function__ = torch._C._disabled_torch_function_impl

    def __new__(cls, group, requires_grad=True):
        data = torch.zeros(group.tangent_shape, device=group.data.device, dtype=group.data.dtype, requires_grad=True)
        return torch.Tensor._make_subclass(cls, data, requires_grad)

    def __init__(self, group, requires_grad=True):
        self.group = group

    def retr(self):
        return self.group.retr(self)

    def inv(self):
        return self.retr().inv()


def pose7_identity(device):
    return torch.tensor([[0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0]], device=device)


class StubGaussians:
    """The attributes of `HTGaussianModel` the render wrapper reads (gaussian_model_ht.py:67-90,128-188), on top of an object that
    holds the six raw tensors and the optimizer (train_step.GaussianParams)."""

    def __init__(self, params):
        self._p = params
        self.rotate_xyz = False
        self.rotate_seq = False
        self.seq_idx = 0
        self.P = None
        N, dev = params._xyz.shape[0], params._xyz.device
        self.max_radii2D = torch.zeros(N, device=dev)
        self.xyz_gradient_accum = torch.zeros(N, 1, device=dev)
        self.denom = torch.zeros(N, 1, device=dev)

    _xyz = property(lambda s: s._p._xyz)
    _features_dc = property(lambda s: s._p._features_dc)
    _features_rest = property(lambda s: s._p._features_rest)
    _opacity = property(lambda s: s._p._opacity)
    _scaling = property(lambda s: s._p._scaling)
    _rotation = property(lambda s: s._p._rotation)
    active_sh_degree = property(lambda s: s._p.active_sh_degree)
    max_sh_degree = property(lambda s: s._p.max_sh_degree)
    optimizer = property(lambda s: s._p.optimizer)

    @property
    def get_xyz(self):                       # gaussian_model_ht.py:135-148
        if self.rotate_xyz:
            return self.P[0].retr().act(self._xyz.clone())
        if self.rotate_seq:
            return self.P[self.seq_idx].retr().act(self._xyz.clone())
        return self._xyz

    get_scaling = property(lambda s: torch.exp(s._scaling))
    get_rotation = property(lambda s: torch.nn.functional.normalize(s._rotation))
    get_opacity = property(lambda s: torch.sigmoid(s._opacity))
    get_features = property(lambda s: torch.cat((s._features_dc, s._features_rest), dim=1))

    def add_densification_stats(self, viewspace_point_tensor, update_filter):      # gaussian_model_ht.py:718-721, verbatim semantics
        self.xyz_gradient_accum[update_filter] += torch.norm(viewspace_point_tensor.grad[update_filter, :2], dim=-1, keepdim=True)
        self.denom[update_filter] += 1


class StubRender:
    """`CF3DGS_Render` as far as `render` reads it: `.gaussians`, `.bg_color` (gaussian_model_ht.py:726-741).  `render` is the
    ORIGINAL wrapper's sequence of calls restated (activated tensors + cat -> GaussianRasterizer -> clamp), i.e. the unpatched route
    against which the patched one is compared."""

    def __init__(self, params, bg=(0.0, 0.0, 0.0)):
        self.gaussians = StubGaussians(params)
        self.bg_color = torch.tensor(bg, dtype=torch.float32, device=params._xyz.device)

    def render(self, viewpoint_camera, scaling_modifier=1.0, invert_bg_color=False, override_color=None,
               compute_cov3D_python=False, convert_SHs_python=False):
        from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer
        g = self.gaussians
        xyz = g.get_xyz
        screenspace_points = torch.zeros_like(xyz, requires_grad=True) + 0       # :800-808
        screenspace_points.retain_grad()
        rs = GaussianRasterizationSettings(
            image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
            tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
            bg=self.bg_color if not invert_bg_color else 1 - self.bg_color, scale_modifier=scaling_modifier,
            viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
            sh_degree=g.active_sh_degree, campos=viewpoint_camera.camera_center, prefiltered=False, debug=False)
        out = GaussianRasterizer(raster_settings=rs)(
            means3D=xyz, means2D=screenspace_points, shs=None if override_color is not None else g.get_features,
            colors_precomp=override_color, opacities=g.get_opacity, scales=g.get_scaling, rotations=g.get_rotation, cov3D_precomp=None)
        image, radii, depth, alpha = out
        return {"image": image.clamp(0, 1), "depth": depth, "alpha": alpha, "viewspace_points": screenspace_points,
                "visibility_filter": radii > 0, "radii": radii}
