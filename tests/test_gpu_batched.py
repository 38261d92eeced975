"""Batched rendering of independent models (GsrBatch): every model of a batch must come out exactly as if rendered / trained alone."""
import importlib

import numpy as np
import pytest
import torch

import parity

pytestmark = pytest.mark.gpu
ts = importlib.import_module("3dgs_hierarchical_training_amd.train_step")
bt = importlib.import_module("3dgs_hierarchical_training_amd.batched")
raster = importlib.import_module("3dgs_hierarchical_training_amd.rasterizer")
L = importlib.import_module("3dgs_hierarchical_training_amd._lib")
NAMES = ["_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"]


def _scenes(sizes, W, H, deg, posed=True):
    return [parity.syn.make_scene(n, W, H, sh_degree=deg, seed=40 + k, posed=posed) for k, n in enumerate(sizes)]


@pytest.mark.parametrize("sizes,deg", [((5000, 12800, 7001), 3), ((3000, 2999), 0), ((129, 4000, 127, 6000, 2048), 1)],
                         ids=["three-deg3", "two-deg0", "five-small-deg1"])
def test_batched_training_equals_training_each_model_alone(sizes, deg):
    """B models in one store, one launch chain per step (fused Adam + the hand-over of the next step's preprocess), against the same
    models trained one by one: images, radii, depth, alpha, screen-space gradients, parameters and Adam moments are EQUAL bit for
    bit, for models of different sizes (padded to 128-Gaussian blocks), different cameras and different targets.  (Deterministic
    backward: with float atomics two runs of the SAME path already differ in the last bits.)"""
    lib = L.load()
    dev = torch.device("cuda:0")
    W, H = 330, 250            # ragged tile grid: the last tile row / column of every image is partial
    scenes = _scenes(sizes, W, H, deg)
    B = len(sizes)
    gts = [parity.syn.target_image(W, H, seed=10 + k).to(dev) for k in range(B)]
    # two cameras per model, alternating (the hand-over always prepares the other one)
    cams = []
    for k, sc in enumerate(scenes):
        alt = parity.syn.make_scene(8, W, H, sh_degree=deg, seed=70 + k, posed=True)
        sc2 = dict(sc)
        for key in ("viewmatrix", "projmatrix", "campos"):
            sc2[key] = alt[key]
        cams.append([ts.make_settings(sc, dev, deg), ts.make_settings(sc2, dev, deg)])
    assert lib.gsr_set_option(b"deterministic_backward", 1) == 0
    try:
        singles = [ts.GaussianParams(sc, dev) for sc in scenes]
        batch = bt.BatchedGaussianParams(scenes, dev)
        assert batch.num_points % 128 == 0 and batch.first_block[-1] * 128 == batch.num_points
        bviews = [bt.batch_settings([cams[k][v] for k in range(B)], dev) for v in range(2)]
        gt_stack = torch.stack(gts)
        for it in range(5):
            v = it % 2
            pk = ts.train_step(batch, bviews[v], gt_stack, next_settings=bviews[1 - v])
            assert pk["raw_image"].shape == (B, 3, H, W) and pk["depth"].shape == (B, 1, H, W)
            if it:
                assert getattr(batch, "_prepared", None) is not None
            for k in range(B):
                ps = ts.train_step(singles[k], cams[k][v], gts[k], next_settings=cams[k][1 - v])
                rows = batch.model_rows(k)
                assert torch.equal(pk["raw_image"][k], ps["raw_image"]), (it, k)
                assert torch.equal(pk["depth"][k], ps["depth"]) and torch.equal(pk["alpha"][k], ps["alpha"]), (it, k)
                assert torch.equal(pk["radii"][rows], ps["radii"]), (it, k)
                assert torch.equal(pk["viewspace_points"].grad[rows], ps["viewspace_points"].grad), (it, k)
                for name in NAMES:
                    assert torch.equal(getattr(batch, name).detach()[rows], getattr(singles[k], name).detach()), (it, k, name)
                for gb, gs in zip(batch.optimizer.param_groups, singles[k].optimizer.param_groups):
                    sb, ss = batch.optimizer.state[gb["params"][0]], singles[k].optimizer.state[gs["params"][0]]
                    assert torch.equal(sb["exp_avg"][rows], ss["exp_avg"]) and torch.equal(sb["exp_avg_sq"][rows], ss["exp_avg_sq"]), (it, k, gb["name"])
        # padding Gaussians never moved and never drew anything
        pad = torch.ones(batch.num_points, dtype=torch.bool, device=dev)
        for k in range(B):
            pad[batch.model_rows(k)] = False
        if bool(pad.any()):
            assert int(pk["radii"][pad].abs().sum()) == 0
            assert float(batch.optimizer.state[batch._xyz]["exp_avg"][pad].abs().sum()) == 0.0
    finally:
        lib.gsr_set_option(b"deterministic_backward", 0)


def test_batched_pose_gradients_equal_single_renders():
    """Frozen models under per-model pose transforms (stage A's pose fit: `points_transform` [B,3,4]): images and dL/d(transform) of
    the batch equal the single renders (the camera-gradient partials are reduced per model over its own blocks)."""
    lib = L.load()
    dev = torch.device("cuda:0")
    W, H = 256, 192
    sizes = (6000, 3001, 9000)
    scenes = _scenes(sizes, W, H, 3, posed=False)
    B = len(sizes)
    pose_mod = importlib.import_module("3dgs_hierarchical_training_amd.pose")
    Ms = [pose_mod.se3_exp(torch.tensor(v))[:3].contiguous().to(dev) for v in
          ([0.0] * 6, [0.02, -0.01, 0.015, 0.004, -0.003, 0.002], [-0.015, 0.01, 0.02, -0.002, 0.004, 0.001])]
    ident = ts.make_settings(scenes[0], dev, 3)
    w = torch.rand(B, 3, H, W, device=dev)
    assert lib.gsr_set_option(b"deterministic_backward", 1) == 0
    try:
        batch = bt.BatchedGaussianParams(scenes, dev, optimizer="torch")
        raw = batch.raw()
        bset = bt.batch_settings([ident] * B, dev)
        Mb = torch.stack(Ms).requires_grad_(True)
        m2d = torch.zeros_like(raw["_xyz"])
        img = raster.rasterize_gaussians_raw(raw["_xyz"], m2d, raw["_features_dc"], raw["_features_rest"], raw["_opacity"], raw["_scaling"],
                                             raw["_rotation"], bset, points_transform=Mb, batch_first_block=batch.first_block)[0]
        (img * w).sum().backward()
        for k in range(B):
            r = batch.model_raw(k)
            Mk = Ms[k].clone().requires_grad_(True)
            m2 = torch.zeros_like(r["_xyz"])
            one = raster.rasterize_gaussians_raw(r["_xyz"], m2, r["_features_dc"], r["_features_rest"], r["_opacity"], r["_scaling"],
                                                 r["_rotation"], ident, points_transform=Mk)[0]
            (one * w[k]).sum().backward()
            assert torch.equal(img[k].detach(), one.detach()), k
            assert torch.equal(Mb.grad[k], Mk.grad), (k, (Mb.grad[k] - Mk.grad).abs().max().item())
    finally:
        lib.gsr_set_option(b"deterministic_backward", 0)
