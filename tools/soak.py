"""Stability soak: training steps over six alternating views with densification surgery (clone 3 %, prune ~3 %, opacity reset)
every 500 / 3000 steps, checking for non-finite parameters.  Needs a GPU:  gpurun -- 'python tools/soak.py [steps] [gaussians]'
(round 1: 12 000 steps at 300 k in 9 s and 40 000 steps at 1 M -> 2.9 M Gaussians in 73 s, no non-finite value, no hang;
round 2 runs the loop with the hand-over to the next view, as run_segments.py does; round 3: the model starts at SH degree 0 and is
raised every 1 000 steps with the hand-over kept across the change, and a real `Densifier` collects its statistics inside the backward
kernel and densifies / prunes every 500 steps on top of the random surgery; round 4, with the direct binning and the persistent backward
blend: 40 000 steps at 1 M in 72 s, 12 000 steps at 300 k in 9.0 s, no non-finite value, no hang; round 5, with the early instance count,
the sub-ulp pixel remainders and the prefetched moment stream: 40 000 steps at 1 M (741 k Gaussians at the end) in 94 s on a slower box,
12 000 steps at 300 k in 8.3 s; same-box A/B against the round-4 tree: the same trajectory -- Gaussian counts within 0.3 %, losses to
three digits -- 8 000 steps at 1 M in 7.0 s against 7.6 s, 6 000 at 300 k in 4.0 s against 4.7 s: profiles/r05_soak.txt)."""
import sys, time, importlib, torch
sys.path.insert(0, '.')   # run from the repository root
syn = importlib.import_module('3dgs_hierarchical_training_amd.synthetic')
ts = importlib.import_module('3dgs_hierarchical_training_amd.train_step')
dev = torch.device('cuda:0')
N, W, H = int(sys.argv[2]) if len(sys.argv) > 2 else 300000, 980, 545
scene = syn.make_scene(N, W, H, sh_degree=3, seed=3)
gen = torch.Generator().manual_seed(0)
views = []
for k in range(6):
    c = syn.make_camera(W, H, R=syn.random_rotation(gen, 0.2), t=0.2 * torch.randn(3, generator=gen))
    s = dict(scene); s.update(c); views.append(ts.make_settings(s, dev, 3))
gts = [syn.target_image(W, H, seed=10 + k).to(dev) * 0.5 + 0.25 for k in range(6)]
params = ts.GaussianParams(scene, dev)
params.active_sh_degree = 0
dm = importlib.import_module('3dgs_hierarchical_training_amd.densify')
den = dm.Densifier(params, scene_extent=5.0, cfg=dm.DensifyConfig(densify_from_iter=400, densification_interval=500, densify_grad_threshold=4e-4,
                                                                 opacity_reset_interval=10 ** 9, max_points=4 * N), seed=1)
t0 = time.time()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
for it in range(steps):
    v = it % 6
    up = (it + 1) % 1000 == 0 and params.active_sh_degree < 3
    pkg = ts.train_step(params, views[v], gts[v], next_settings=views[(it + 1) % 6], densifier=den, iteration=it + 1,
                        next_sh_degree=params.active_sh_degree + 1 if up else None)   # prepare in backward: the production loop
    if up:
        params.oneup_sh_degree()
    if it % 500 == 499:
        n = params._xyz.shape[0]
        # densify: clone 3 % (random), prune 3 % (lowest opacity), like the reference's cadence
        idx = torch.randperm(n, device=dev)[: n // 33]
        new = {g["name"]: g["params"][0].detach()[idx].clone() for g in params.optimizer.param_groups}
        params.densification_postfix(new)
        den.reset_stats()
        op = params.get_opacity.detach().squeeze(1)
        thr = torch.quantile(op[torch.randperm(op.numel(), device=dev)[:100000]], 0.03)
        params.prune_points(op < thr)
        den.reset_stats()
        if it % 3000 == 2999:
            params.reset_opacity()
        l = float(pkg["loss"])
        bad = any(not torch.isfinite(g["params"][0]).all() for g in params.optimizer.param_groups)
        import resource
        print(it + 1, "N", params._xyz.shape[0], "SH degree", params.active_sh_degree, "loss %.5f" % l, "nonfinite", bad, "%.1f s" % (time.time() - t0),
              "device reserved %d MiB, host RSS max %d MiB" % (torch.cuda.memory_reserved(dev) >> 20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10), flush=True)
        assert not bad and l == l
torch.cuda.synchronize()
print("done", steps, "steps in %.1f s" % (time.time() - t0))
