"""gsr_autopatch, round 6: the frame poses of the unmodified trainer (lietorch `LieGroupParameter` + its Adam) on the kernels --
the DISPATCH logic, on CPU.  The kernels' numbers are checked on the GPU (tests/test_gpu_autopatch.py, tests/test_gpu_pose.py).

The HIP ops cannot run here, so `gsr_autopatch._ops` is replaced by a recorder whose `pose_matrix` is pose.py's torch statement
of Exp(delta) * base (test infrastructure, the same closed forms tests/test_gpu_pose.py holds the kernel to)."""
import importlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
refstub = importlib.import_module("3dgs_hierarchical_training_amd.refstub")
pose = importlib.import_module("3dgs_hierarchical_training_amd.pose")
pose_opt = importlib.import_module("3dgs_hierarchical_training_amd.pose_opt")
from test_autopatch_render_cpu import _Params        # noqa: E402  (the stub model built from the captured boundary arguments)


class _RecordingOps:
    def __init__(self):
        self.calls = []

    def pose_matrix(self, delta, base):
        self.calls.append(("pose_matrix", delta, base))
        B = torch.cat((base, torch.tensor([[0.0, 0.0, 0.0, 1.0]])), 0)
        return (pose.se3_exp(delta.reshape(6)) @ B)[:3]


@pytest.fixture
def autopatch(monkeypatch):
    import gsr_autopatch
    gsr_autopatch.apply()
    gsr_autopatch._REQUIRE_CUDA = False
    ops = _RecordingOps()
    monkeypatch.setattr(gsr_autopatch, "_ops", lambda: ops)
    gsr_autopatch._test_ops = ops
    yield gsr_autopatch
    gsr_autopatch._REQUIRE_CUDA = True
    gsr_autopatch.remove()


def _lie(pose7=None, delta=None):
    p = refstub.LieGroupParameter(refstub.SE3(torch.tensor([pose7 if pose7 is not None else [0.0, 0, 0, 0, 0, 0, 1]])))
    if delta is not None:
        with torch.no_grad():
            p.copy_(torch.tensor([delta]))
    return p


def test_is_lie_pose_is_exactly_the_lietorch_shape():
    p = _lie()
    assert pose_opt.is_lie_pose(p) and tuple(p.shape) == (1, 6) and type(p.data) is torch.Tensor
    assert not pose_opt.is_lie_pose(torch.zeros(1, 6, requires_grad=True))                 # no group element
    q = torch.zeros(1, 6, requires_grad=True)
    q.group = object()
    assert not pose_opt.is_lie_pose(q)                                                       # a group that is not an SE3

    class SO3:                                                                                # another lietorch group: 4 numbers, 3 tangent
        data = torch.zeros(1, 4)
    q.group = SO3()
    assert not pose_opt.is_lie_pose(q)
    assert not pose_opt.is_lie_pose(_lie().detach())                                         # does not want a gradient
    # the base matrix is the group element's [3,4] matrix, cached until the element is replaced or written
    p = _lie([0.1, -0.2, 0.3, 0.1, 0.2, -0.1, 0.9])
    B = pose_opt.base_matrix(p)
    assert torch.allclose(B, pose.pose7_to_matrix(p.group.data.reshape(7))[:3], atol=1e-7) and pose_opt.base_matrix(p) is B
    p.group = refstub.SE3(torch.tensor([[0.0, 0, 0, 0, 0, 0, 1]]))                          # update_RT_seq's statement (gaussian_model_ht.py:386)
    assert torch.equal(pose_opt.base_matrix(p), torch.eye(4)[:3])
    p.group.data.mul_(2.0)                                                                   # written in place: the version counter moves
    assert torch.allclose(pose_opt.base_matrix(p), torch.cat((torch.eye(3), torch.zeros(3, 1)), 1))   # (q normalised: still the identity rotation)


def test_adam_dispatch_hands_out_the_pose_optimizer_for_the_references_constructions(autopatch):
    """`camera_optimizer[k]` (gaussian_model_ht.py:296-311) and stage A's `training_setup_fix_position(gaussian_rot=False)` (:321-333):
    a group named 'R' over one LieGroupParameter.  With `_rotation` beside it (`gaussian_rot=True`), or a plain tensor, the stock class."""
    p = _lie()
    o = torch.optim.Adam([{'params': [p], 'lr': 1e-3, "name": "R"}], lr=0.0, eps=1e-15)
    assert type(o) is pose_opt.FusedPoseAdam and isinstance(o, torch.optim.Adam)
    assert o.param_groups[0]["params"][0] is p and o.param_groups[0]["lr"] == 1e-3 and o.param_groups[0]["name"] == "R" and o.eps == 1e-15
    for g in o.param_groups:                      # update_learning_rate_camera's statement (:396-405)
        g["lr"] = 5e-4
    rot = torch.zeros(10, 4, requires_grad=True)
    o2 = torch.optim.Adam([{'params': [p], 'lr': 1e-3, "name": "R"}, {'params': [rot], 'lr': 1e-3, "name": "rotation"}], lr=0.0, eps=1e-15)
    assert type(o2) is autopatch._ORIG_ADAM
    assert type(torch.optim.Adam([torch.zeros(1, 6, requires_grad=True)], lr=1e-3)) is autopatch._ORIG_ADAM
    os.environ["GSR_AUTOPATCH_POSE_FUSED"] = "0"
    try:
        assert type(torch.optim.Adam([{'params': [p], 'lr': 1e-3, "name": "R"}], lr=0.0, eps=1e-15)) is autopatch._ORIG_ADAM
    finally:
        del os.environ["GSR_AUTOPATCH_POSE_FUSED"]
    # CPU parameters never reach the kernels silently
    p.grad = torch.ones(1, 6)
    with pytest.raises(RuntimeError, match="ROCm/HIP device"):
        o.step()
    o.zero_grad(set_to_none=True)
    assert p.grad is None
    o.step()                                      # nothing to step: no error, no state
    assert o.state == {}


def test_pose_optimizer_state_dict_is_torchs_layout(autopatch):
    """capture / restore (gaussian_model_ht.py:102,124) and a checkpoint written by the stock class: same keys, same meaning."""
    p = _lie()
    o = torch.optim.Adam([{'params': [p], 'lr': 1e-3, "name": "R"}], lr=0.0, eps=1e-15)
    stock = autopatch._ORIG_ADAM([{'params': [p], 'lr': 1e-3, "name": "R"}], lr=0.0, eps=1e-15)
    p.grad = torch.full((1, 6), 0.5)
    stock.step()
    sd = stock.state_dict()
    o.load_state_dict(sd)
    st = o.state[p]
    assert st["step"] == 1 and torch.equal(st["exp_avg"], stock.state[p]["exp_avg"]) and torch.equal(st["exp_avg_sq"], stock.state[p]["exp_avg_sq"])
    back = o.state_dict()
    assert set(back) == {"state", "param_groups"} and back["param_groups"][0]["params"] == [0] and back["param_groups"][0]["name"] == "R"
    assert float(back["state"][0]["step"]) == 1.0
    stock2 = autopatch._ORIG_ADAM([{'params': [p], 'lr': 1e-3, "name": "R"}], lr=0.0, eps=1e-15)
    stock2.load_state_dict(back)                  # ... and the stock class reads what this one writes
    assert torch.equal(stock2.state[p]["exp_avg"], st["exp_avg"])


def _stub_render(d):
    p = _Params(d)
    r = refstub.StubRender(p, bg=tuple(d["kernel_st_bg"]))
    W, H = int(d["kernel_st_image_width"]), int(d["kernel_st_image_height"])
    cam = refstub.StubCamera(W, H, float(d["kernel_st_tanfovx"]), float(d["kernel_st_tanfovy"]), torch.eye(4), torch.from_numpy(d["kernel_st_projmatrix"].copy()),
                             torch.zeros(3), uid=3)
    return p, r, cam, W, H


def _fake_raster(rec, H, W, weight):
    def fake(means3D, means2D, f_dc, f_rest, opacity, scaling, rotation, settings, **kw):
        rec.update(t=(means3D, means2D), kw=kw)
        xf = kw.get("points_transform")
        z = means3D.sum() * 0 + (0 if xf is None else (xf[:3] * weight).sum())
        out = (torch.zeros(3, H, W) + z, torch.ones(means3D.shape[0], dtype=torch.int32), torch.zeros(1, H, W), torch.zeros(1, H, W))
        if kw.get("extras"):
            out = out + (out[0].clamp(0, 1), (out[1] > 0).to(torch.uint8))
        return out
    return fake


@pytest.mark.parametrize("mode", ["rotate_seq", "rotate_xyz"])
def test_pose_render_goes_through_one_node_and_leaves_the_gradient_on_the_parameter(autopatch, mode):
    """`get_xyz`'s `P[k].retr().act(xyz)` (gaussian_model_ht.py:135-148) on the patched render: the means stay the raw `_xyz`, the
    transform is the [3,4] tensor of ONE `pose_matrix` node over P[k] (no retr() chain), it equals lietorch's `retr().matrix()`, and
    its backward leaves on `P[k].grad` what the chain would have left."""
    d = np.load(os.path.join(GOLD, "boundary_args.npz"))
    p, r, cam, W, H = _stub_render(d)
    g = r.gaussians
    poses = [_lie([0.1, -0.2, 0.3, 0.1, 0.2, -0.1, 0.9], [0.01, -0.02, 0.03, 0.02, -0.01, 0.015]), _lie([0.0, 0.1, 0.0, 0.0, 0.1, 0.0, 1.0], [0.0, 0.01, 0.0, -0.02, 0.0, 0.01])]
    if mode == "rotate_seq":
        g.P, g.rotate_seq, g.seq_idx = poses, True, 1
        live, other = poses[1], poses[0]
    else:
        g.P, g.rotate_xyz = poses[:1], True
        live, other = poses[0], poses[1]
    R = importlib.import_module("3dgs_hierarchical_training_amd.rasterizer")
    rec, weight = {}, torch.arange(12.0).reshape(3, 4) * 0.1 - 0.4
    orig, R.rasterize_gaussians_raw = R.rasterize_gaussians_raw, _fake_raster(rec, H, W, weight)
    try:
        pkg = autopatch.render_fused(r, cam)
    finally:
        R.rasterize_gaussians_raw = orig
    ops = autopatch._test_ops
    assert len(ops.calls) == 1 and ops.calls[0][1] is live and tuple(ops.calls[0][2].shape) == (3, 4)
    xf = rec["kw"]["points_transform"]
    assert tuple(xf.shape) == (3, 4) and rec["t"][0] is p._xyz
    want = live.retr().matrix().reshape(4, 4)[:3]                 # lietorch's statement of the same element
    assert torch.allclose(xf, want, atol=1e-6)
    pkg["image"][0, 0, 0].backward()
    ref = torch.autograd.grad((want * weight).sum(), live)[0]
    assert live.grad is not None and tuple(live.grad.shape) == (1, 6) and torch.allclose(live.grad, ref, rtol=1e-5, atol=1e-7)
    assert other.grad is None
    if mode == "rotate_seq":                                      # the frame id carries the pose slot (view-cost cache key)
        assert rec["kw"]["view_id"] == 1 + 2 * 3 + 1000003 * 2


def test_other_pose_objects_keep_the_original_statement(autopatch):
    """`rotate_xyz_inverse` (retr().inv()), a pose object that is not a lietorch SE3 parameter, and GSR_AUTOPATCH_POSE_FUSED=0 all
    evaluate `retr()` as the original `get_xyz` does: a [4,4] matrix with its own autograd chain, no pose node."""
    d = np.load(os.path.join(GOLD, "boundary_args.npz"))
    p, r, cam, W, H = _stub_render(d)
    g = r.gaussians
    R = importlib.import_module("3dgs_hierarchical_training_amd.rasterizer")
    rec = {}
    orig, R.rasterize_gaussians_raw = R.rasterize_gaussians_raw, _fake_raster(rec, H, W, torch.ones(3, 4))
    ops = autopatch._test_ops
    try:
        lie = _lie([0.1, -0.2, 0.3, 0.1, 0.2, -0.1, 0.9], [0.01, -0.02, 0.03, 0.02, -0.01, 0.015])
        g.P, g.rotate_xyz_inverse = [lie], True
        pkg = autopatch.render_fused(r, cam)
        assert not ops.calls and tuple(rec["kw"]["points_transform"].shape) == (4, 4)
        assert torch.allclose(rec["kw"]["points_transform"], torch.linalg.inv(lie.retr().matrix().reshape(4, 4)), atol=1e-6)
        pkg["image"][0, 0, 0].backward()
        assert lie.grad is not None
        g.rotate_xyz_inverse = False

        class _T:
            def __init__(self, M):
                self.M = M

            def matrix(self):
                return self.M[None]

        class _P:
            def __init__(self):
                self.t = torch.zeros(3, requires_grad=True)

            def retr(self):
                return _T(torch.eye(4) + torch.cat([torch.cat([torch.zeros(3, 3), self.t[:, None]], 1), torch.zeros(1, 4)], 0))
        g.P, g.rotate_xyz = [_P()], True
        autopatch.render_fused(r, cam)["image"][0, 0, 0].backward()
        assert not ops.calls and g.P[0].t.grad is not None
        os.environ["GSR_AUTOPATCH_POSE_FUSED"] = "0"
        try:
            g.P = [_lie()]
            autopatch.render_fused(r, cam)
            assert not ops.calls and tuple(rec["kw"]["points_transform"].shape) == (4, 4)
        finally:
            del os.environ["GSR_AUTOPATCH_POSE_FUSED"]
        g.P = [_lie()]
        autopatch.render_fused(r, cam)
        assert len(ops.calls) == 1
    finally:
        R.rasterize_gaussians_raw = orig


def test_zero_points_buffer_is_not_shared_after_a_write(autopatch):
    """the screen-space leaf of every render aliases one cached zero buffer; an in-place write under no_grad (the version
    counter every detach() of the buffer shares moves) must not reach the next render's leaf, and sizes other than the two most
    recent are dropped.  (A write through `.data` has a version counter of its own and cannot be seen: documented in render_fused.)"""
    d = np.load(os.path.join(GOLD, "boundary_args.npz"))
    p, r, cam, W, H = _stub_render(d)
    R = importlib.import_module("3dgs_hierarchical_training_amd.rasterizer")
    rec = {}
    orig, R.rasterize_gaussians_raw = R.rasterize_gaussians_raw, _fake_raster(rec, H, W, torch.ones(3, 4))
    try:
        a = autopatch.render_fused(r, cam)["viewspace_points"]
        b = autopatch.render_fused(r, cam)["viewspace_points"]
        assert a is not b and a.data_ptr() == b.data_ptr() and a.is_leaf and b.is_leaf
        with torch.no_grad():
            a.add_(1.0)                            # somebody writes into a leaf's storage
        c = autopatch.render_fused(r, cam)["viewspace_points"]
        assert c.data_ptr() != a.data_ptr() and float(c.abs().max()) == 0.0
        assert len(autopatch._ZERO_POINTS) <= 2
    finally:
        R.rasterize_gaussians_raw = orig


@pytest.mark.skipif(not os.path.isdir("/root/reference/scene"), reason="authoring container only: imports the real reference classes")
def test_pose_route_on_the_real_reference_classes():
    """The REAL `HTGaussianModel.init_RT_seq` / `training_setup(fit_pose=True)` / `training_setup_fix_position` / `update_RT_seq` /
    `get_RT` and `CF3DGS_Render.render` (imported under the CPU shim, lietorch replaced by refstub's stand-ins of its public API)
    with `import gsr_autopatch` first: the optimizers they build are FusedPoseAdam objects, their renders take the pose node."""
    out = subprocess.run([sys.executable, os.path.join(HERE, "helpers", "ref_pose_driver.py")], capture_output=True, text=True, timeout=600)
    line = next((l for l in out.stdout.splitlines() if l.startswith("RESULT ")), None)
    assert line is not None, out.stdout[-2000:] + out.stderr[-4000:]
    res = json.loads(line[7:])
    assert res["camera_optimizers"] == ["FusedPoseAdam"] * 3 and res["camera_optimizer_is_adam"]
    assert res["model_optimizer"] == "FusedAdam" or res["model_optimizer"] == "Adam"         # (CPU tensors: the six-group optimizer stays stock)
    assert res["fix_position_optimizer"] == "FusedPoseAdam" and res["fix_position_with_rotation"] == "Adam"
    assert res["seq_transform_shape"] == [3, 4] and res["seq_node_calls"] == 1 and res["seq_node_param_is_P2"]
    assert res["seq_transform_error"] < 1e-6 and res["seq_grad_error"] < 1e-5 and res["seq_other_grads_none"]
    assert res["get_RT_matches"] < 1e-6
    assert res["after_update_RT_seq_error"] < 1e-6
    assert res["lr_statement_ok"]
