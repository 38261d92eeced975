"""Where a densification iteration of the C3 workload (1 M Gaussians @1920x1080, every 100 steps) spends its time: per call of
densification_postfix / prune_points / densify_and_prune / optimizer.step, with the caching allocator's device allocations.
On a box with a cold page cache the first clone / split pay torch's kernel loads and the rocBLAS initialisation (bmm): 570 + 14 + 266 ms
for the three densifications of a 300-step run; bench.py warms that path on a throw-away model (warm_densify).  gpurun -- python tools/densify_probe.py"""
import importlib, sys, os, time, torch
sys.path.insert(0, "/root/repo")
syn = importlib.import_module("3dgs_hierarchical_training_amd.synthetic")
ts = importlib.import_module("3dgs_hierarchical_training_amd.train_step")
dm = importlib.import_module("3dgs_hierarchical_training_amd.densify")
dev = torch.device("cuda:0")
N, W, H, deg = 1_000_000, 1920, 1080, 3
scene = syn.make_scene(N, W, H, sh_degree=deg, seed=0)
gt = syn.target_image(W, H, seed=1).to(dev)
p = ts.GaussianParams(scene, dev)
st = ts.make_settings(scene, dev, deg)
den = dm.Densifier(p, scene_extent=5.0, cfg=dm.DensifyConfig(densify_from_iter=0, densification_interval=100, densify_grad_threshold=2e-4, opacity_reset_interval=10 ** 9, max_points=2 * N))
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); s0 = torch.cuda.memory_stats()
        r = f(*a, **k)
        torch.cuda.synchronize(); s1 = torch.cuda.memory_stats()
        print(f"   {name}: {1e3 * (time.perf_counter() - t0):8.2f} ms  device allocs +{s1['num_device_alloc'] - s0['num_device_alloc']} frees +{s1['num_device_free'] - s0['num_device_free']} retries +{s1['num_alloc_retries'] - s0['num_alloc_retries']} reserved {s1['reserved_bytes.all.current'] / 2**30:.2f} GiB")
        return r
    setattr(obj, name, g)
wrap(p, "densification_postfix"); wrap(p, "prune_points"); wrap(den, "densify_and_prune"); wrap(p.optimizer, "step")
for i in range(1, 305):
    verbose = i % 100 == 0
    if not verbose:
        # silence the wrappers outside densify iterations
        import io, contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            ts.train_step(p, st, gt, densifier=den, iteration=i, next_settings=st)
    else:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        print("step", i, "N", p.num_points)
        ts.train_step(p, st, gt, densifier=den, iteration=i, next_settings=st)
        torch.cuda.synchronize(); print("   whole step", 1e3 * (time.perf_counter() - t0), "ms")
