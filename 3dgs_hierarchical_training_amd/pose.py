"""SE(3) pose parametrisation for the fused pose action ("next" row f-4 of SURVEY.md section 8).

Under pose fitting the reference moves every Gaussian centre through a lietorch group element before rendering,
`xyz = self.P[k].retr().act(self._xyz.clone())` (/root/reference/scene/gaussian_model_ht.py:135-148), where `P[k]` is
a `LieGroupParameter(SE3(pose7))` (:355-372): a trainable 6-vector delta in the tangent space at a fixed group element
G, `retr()` = Exp(delta) * G, pose7 = (tx, ty, tz, qx, qy, qz, qw), tangent order (translation tau, rotation phi).
That is an N x 3 clone + an N x 3 act (and their backward) per step, outside the rasterizer.

Here the 3x4 matrix of Exp(delta) * G is built with a handful of differentiable torch ops on 6 + 7 numbers and handed
to the kernels as `points_transform` (rasterizer.py): the act runs in-kernel on the means as they are loaded, and
dL/d(matrix) comes back from the backward kernel, so autograd delivers dL/d(delta) without touching N-sized tensors.
lietorch itself is not needed (and is not installed here); the maps below are the textbook closed forms.
"""
import torch


def _hat(v: torch.Tensor) -> torch.Tensor:
    z = torch.zeros((), dtype=v.dtype, device=v.device)
    return torch.stack((torch.stack((z, -v[2], v[1])), torch.stack((v[2], z, -v[0])), torch.stack((-v[1], v[0], z))))


def so3_exp_and_v(phi: torch.Tensor):
    """Rodrigues rotation R = exp(phi^) and the left Jacobian V with t = V tau, both [3,3]; series near 0."""
    th2 = (phi * phi).sum()
    small = th2 < 1e-8
    th2s = torch.where(small, torch.ones_like(th2), th2)
    th = torch.sqrt(th2s)
    a = torch.where(small, 1.0 - th2 / 6.0, torch.sin(th) / th)                    # sin t / t
    b = torch.where(small, 0.5 - th2 / 24.0, (1.0 - torch.cos(th)) / th2s)         # (1 - cos t) / t^2
    c = torch.where(small, 1.0 / 6.0 - th2 / 120.0, (th - torch.sin(th)) / (th2s * th))   # (t - sin t) / t^3
    K = _hat(phi)
    K2 = K @ K
    eye = torch.eye(3, dtype=phi.dtype, device=phi.device)
    return eye + a * K + b * K2, eye + b * K + c * K2


def se3_exp(delta: torch.Tensor) -> torch.Tensor:
    """Exp of a tangent vector (tau[3], phi[3]) -> [4,4]."""
    R, V = so3_exp_and_v(delta[3:])
    top = torch.cat((R, (V @ delta[:3]).unsqueeze(1)), dim=1)
    return torch.cat((top, torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=delta.dtype, device=delta.device)), dim=0)


def quat_xyzw_to_matrix(q: torch.Tensor) -> torch.Tensor:
    q = q / q.norm()
    x, y, z, w = q[0], q[1], q[2], q[3]
    return torch.stack((
        torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w))),
        torch.stack((2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w))),
        torch.stack((2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)))))


def pose7_to_matrix(pose7: torch.Tensor) -> torch.Tensor:
    """(tx, ty, tz, qx, qy, qz, qw) -> [4,4]."""
    top = torch.cat((quat_xyzw_to_matrix(pose7[3:]), pose7[:3].unsqueeze(1)), dim=1)
    return torch.cat((top, torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=pose7.dtype, device=pose7.device)), dim=0)


def retr_matrix(delta: torch.Tensor, pose7: torch.Tensor) -> torch.Tensor:
    """Matrix of Exp(delta) * G -- what `LieGroupParameter.retr()` evaluates to."""
    return se3_exp(delta) @ pose7_to_matrix(pose7)


def act(matrix: torch.Tensor, xyz: torch.Tensor) -> torch.Tensor:
    """R p + t on [N,3] points: the torch statement of what the kernels do with `points_transform`."""
    return xyz @ matrix[:3, :3].t() + matrix[:3, 3]


def so3_log(R: torch.Tensor) -> torch.Tensor:
    """Rotation vector phi with exp(phi^) = R (angle in [0, pi)); series near 0."""
    cos = ((R[0, 0] + R[1, 1] + R[2, 2] - 1.0) * 0.5).clamp(-1.0, 1.0)
    th = torch.acos(cos)
    w = torch.stack((R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1])) * 0.5      # sin(th) * axis
    small = th < 1e-4
    ths = torch.where(small, torch.ones_like(th), th)
    k = torch.where(small, 1.0 + th * th / 6.0, ths / torch.sin(ths))
    return k * w


def se3_log(M: torch.Tensor) -> torch.Tensor:
    """Inverse of se3_exp: [4,4] rigid transform -> tangent (tau[3], phi[3])."""
    phi = so3_log(M[:3, :3])
    _, V = so3_exp_and_v(phi)
    tau = torch.linalg.solve(V, M[:3, 3])
    return torch.cat((tau, phi))


def interpolate_pose(pose0: torch.Tensor, pose1: torch.Tensor, alpha: float) -> torch.Tensor:
    """Virtual view between two poses, pose0 * Exp(alpha * Log(pose0^-1 * pose1)) -- what `get_virtual_view`
    (/root/reference/trainer/ht3dgs_trainer.py:462-479) evaluates with lietorch; alpha = 0 / 1 give pose0 / pose1."""
    assert 0.0 <= alpha <= 1.0
    p0, p1 = pose0.double(), pose1.double()
    return (p0 @ se3_exp(alpha * se3_log(torch.linalg.inv(p0) @ p1))).to(pose0.dtype)
