"""Deterministic synthetic scenes ("syn-N", SURVEY.md section 8d / BASELINE.md section 2).

The camera follows the reference's co3d branch (/root/reference/scene/cameras.py:76-98): the
world-to-view matrix is stored TRANSPOSED (so it is column-major when read linearly), the projection
is the OpenGL-style matrix built from intrinsics, also transposed, and
full_proj = view_T @ proj_T.  FoVx default = arguments/full/Tanks/Francis.yml:46.
Everything is generated on the CPU from a seeded torch.Generator and moved by the caller.
"""
import math
from typing import Dict, Optional

import torch

FOVX_FRANCIS = 1.3541787529604106


def make_camera(W: int, H: int, fovx: float = FOVX_FRANCIS, R: Optional[torch.Tensor] = None,
                t: Optional[torch.Tensor] = None, znear: float = 0.01, zfar: float = 100.0) -> Dict:
    """Pinhole camera with fx = fy, principal point at the centre.  R,t = world-to-camera."""
    fx = 0.5 * W / math.tan(0.5 * fovx)
    fy = fx
    cx, cy = W / 2.0, H / 2.0
    w2c = torch.eye(4, dtype=torch.float32)
    if R is not None:
        w2c[:3, :3] = R.float()
    if t is not None:
        w2c[:3, 3] = t.float()
    view_T = w2c.t().contiguous()
    proj = torch.tensor([[2 * fx / W, 0.0, -(W - 2 * cx) / W, 0.0],
                         [0.0, 2 * fy / H, -(H - 2 * cy) / H, 0.0],
                         [0.0, 0.0, zfar / (zfar - znear), -(zfar * znear) / (zfar - znear)],
                         [0.0, 0.0, 1.0, 0.0]], dtype=torch.float32)
    proj_T = proj.t().contiguous()
    full = (view_T.unsqueeze(0).bmm(proj_T.unsqueeze(0))).squeeze(0).contiguous()
    campos = view_T.inverse()[3, :3].contiguous()
    return dict(image_width=W, image_height=H, tanfovx=math.tan(0.5 * fovx), tanfovy=0.5 * H / fy,
                viewmatrix=view_T, projmatrix=full, campos=campos, fx=fx, fy=fy)


def random_rotation(gen: torch.Generator, max_angle: float = 0.3) -> torch.Tensor:
    axis = torch.randn(3, generator=gen)
    axis = axis / axis.norm()
    ang = (torch.rand(1, generator=gen).item() * 2 - 1) * max_angle
    K = torch.tensor([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return torch.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * (K @ K)


def make_scene(N: int, W: int, H: int, sh_degree: int = 3, seed: int = 0, fovx: float = FOVX_FRANCIS,
               sigma_px: float = 3.0, posed: bool = False, frac_behind: float = 0.02, clustered: bool = False) -> Dict:
    """syn-N: z~U[1,10] filling the frustum with 5% overscan, `frac_behind` of the points near/behind the
    near plane, ~sigma_px anisotropic Gaussians, random unit quaternions, opacity = sigmoid(N(0,2)),
    SH dc ~ 0.5 N(0,1), rest ~ 0.1 N(0,1).  With posed=True the same cloud is seen from a rotated and
    translated camera (exercises the matrix conventions).  clustered=True is the load-imbalance variant: screen positions
    ~ N(centre, 0.13 of the frame) and low opacities sigmoid(N(-2.5, 1)), so a few hundred tiles carry long,
    non-saturating lists (what real scenes look like early in training)."""
    g = torch.Generator().manual_seed(seed)
    if posed:
        Rm = random_rotation(g)
        tv = torch.randn(3, generator=g) * 0.2
        cam = make_camera(W, H, fovx, Rm, tv)
    else:
        cam = make_camera(W, H, fovx)
    tfx, tfy = cam["tanfovx"], cam["tanfovy"]
    z = 1 + 9 * torch.rand(N, generator=g)
    nb = int(N * frac_behind)
    if nb:
        z[:nb] = -1 + 1.2 * torch.rand(nb, generator=g)
    u, v = torch.rand(N, generator=g), torch.rand(N, generator=g)
    if clustered:
        u = (0.5 + 0.13 * torch.randn(N, generator=g)).clamp(-0.02, 1.02)
        v = (0.5 + 0.13 * torch.randn(N, generator=g)).clamp(-0.02, 1.02)
    pc = torch.stack([(2 * u - 1) * 1.05 * tfx * z, (2 * v - 1) * 1.05 * tfy * z, z], 1)
    # camera-space points -> world: p_w = R^T (p_c - t)
    w2c = cam["viewmatrix"].t()
    means = (pc - w2c[:3, 3][None]) @ w2c[:3, :3]
    perm = torch.randperm(N, generator=g)
    means = means[perm].contiguous()
    zc = z[perm].abs().clamp_min(0.3)
    scales = (sigma_px * zc / cam["fx"])[:, None] * torch.exp(0.5 * torch.randn(N, 3, generator=g))
    q = torch.randn(N, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    op = torch.sigmoid(2.0 * torch.randn(N, 1, generator=g))
    if clustered:
        op = torch.sigmoid(-2.5 + torch.randn(N, 1, generator=g))
    M = 16
    shs = torch.zeros(N, M, 3)
    shs[:, 0] = 0.5 * torch.randn(N, 3, generator=g)
    shs[:, 1:] = 0.1 * torch.randn(N, M - 1, 3, generator=g)
    scene = dict(means3D=means.float(), scales=scales.float().contiguous(), rotations=q.float().contiguous(),
                 opacities=op.float().contiguous(), shs=shs.float().contiguous(), sh_degree=sh_degree,
                 bg=torch.zeros(3), scale_modifier=1.0)
    scene.update(cam)
    return scene


def target_image(W: int, H: int, seed: int = 1) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.rand(3, H, W, generator=g)
