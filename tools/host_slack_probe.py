"""Is a train step of this size bound by the device or by the launching thread?  A busy-wait of d microseconds is added to
the host side of every step: if the step time does not move, the host had at least d microseconds of slack (device-bound).
gpurun -- 'python tools/host_slack_probe.py 20000 100000 1000000'"""
import importlib, sys, time
import torch
sys.path.insert(0, '.')
syn = importlib.import_module("3dgs_hierarchical_training_amd.synthetic")
ts = importlib.import_module("3dgs_hierarchical_training_amd.train_step")
dev = torch.device("cuda:0")
W, H = 980, 545
gt = syn.target_image(W, H, seed=1).to(dev)
def spin(us):
    t = time.perf_counter() + us * 1e-6
    while time.perf_counter() < t:
        pass
for N in [int(a) for a in sys.argv[1:]] or [20000, 100000]:
    sc = syn.make_scene(N, W, H, sh_degree=3, seed=0)
    params = ts.GaussianParams(sc, dev)
    st = ts.make_settings(sc, dev, 3)
    for _ in range(10):
        ts.train_step(params, st, gt, next_settings=st)
    out = []
    orig_loss = ts.fused_photometric_loss
    for where in ("before the forward", "between forward and loss"):
        for d in (0, 20, 50, 100, 200):
            torch.cuda.synchronize()
            steps = 200
            t0 = time.perf_counter()
            for _ in range(steps):
                if where == "before the forward":
                    spin(d)
                    ts.train_step(params, st, gt, next_settings=st)
                else:
                    ts.fused_photometric_loss = (lambda *a, **k: (spin(d), orig_loss(*a, **k))[1])
                    ts.train_step(params, st, gt, next_settings=st)
                    ts.fused_photometric_loss = orig_loss
            torch.cuda.synchronize()
            out.append((where, d, 1e3 * (time.perf_counter() - t0) / steps))
    print(f"N = {N}")
    for w, d, ms in out:
        print(f"  +{d:3d} us {w}: {ms:.4f} ms per step")
