"""Stage A's pose fit (train_relative_pose, /root/reference/trainer/ht3dgs_trainer.py:307-333) the way the unmodified trainer runs it under
`import gsr_autopatch`: frozen 130 k-Gaussian model at SH degree 0, `init_RT(None)`, `training_setup_fix_position(gaussian_rot=False)` ->
torch.optim.Adam over the one LieGroupParameter, render through get_xyz's pose, Loss.forward, backward, optimizer.step().  ms per
iteration with the pose on the kernels (pose node + FusedPoseAdam) and with GSR_AUTOPATCH_POSE_FUSED=0 (lietorch's chain stated in torch +
the stock Adam), next to the library's own loop (stage_a.fit_pair: gsr_pose_step)."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gsr_autopatch
ts = importlib.import_module("3dgs_hierarchical_training_amd.train_step")
refstub = importlib.import_module("3dgs_hierarchical_training_amd.refstub")
sequence = importlib.import_module("3dgs_hierarchical_training_amd.sequence")
dev = torch.device("cuda:0")
W, H = 980, 545
seq = sequence.FrameSequence(3, 400_000, W, H, dev, seed=0)
scene = seq.pixel_scene(0, stride=2, seed=0)
tgt1 = seq.target(1)
ident = ts.with_sh_degree(seq.settings_for_pose(torch.eye(4)), 0)
T = seq.true_rel_pose(0, 1)


class _Cfg:
    lambda_dssim, lambda_depth = 0.2, 0.0


class _Loss:
    cfg = _Cfg()


for fused in (1, 0, 1, 0):
    os.environ["GSR_AUTOPATCH_POSE_FUSED"] = str(fused)
    gsr_autopatch.apply()
    try:
        p = ts.GaussianParams(scene, dev, optimizer="torch")
        p.active_sh_degree = 0
        r = refstub.StubRender(p, bg=tuple(float(x) for x in ident.bg.cpu()))
        g = r.gaussians
        g.P = [refstub.LieGroupParameter(refstub.SE3(refstub.pose7_identity(dev)))]
        g.rotate_xyz = True
        opt = torch.optim.Adam([{'params': [g.P[0]], 'lr': 2e-3, "name": "R"}], lr=0.0, eps=1e-15)
        cam = refstub.StubCamera(W, H, ident.tanfovx, ident.tanfovy, ident.viewmatrix, ident.projmatrix, ident.campos, uid=1)

        def step():
            pkg = gsr_autopatch.render_fused(r, cam)
            gsr_autopatch.loss_forward(_Loss(), pkg["image"], tgt1)["loss"].backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 200
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n
        M = torch.eye(4)
        M[:3] = g.P[0].retr().matrix().reshape(4, 4)[:3].detach().cpu()
        print(f"{type(opt).__name__:14s} N={p.num_points} {ms:.3f} ms per pose iteration; pose error after 220 iterations {float((M - T).abs().max()):.2e} (identity guess {float((torch.eye(4) - T).abs().max()):.2e})")
    finally:
        gsr_autopatch.remove()
