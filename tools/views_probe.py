import importlib, sys, time, torch
sys.path.insert(0, '.')
syn = importlib.import_module('3dgs_hierarchical_training_amd.synthetic')
ts = importlib.import_module('3dgs_hierarchical_training_amd.train_step')
host = importlib.import_module("3dgs_hierarchical_training_amd.host"); host.cap_host_threads()
dev = torch.device('cuda:0')
for N in (30000, 300000):
    W, H = 980, 545
    scene = syn.make_scene(N, W, H, sh_degree=3, seed=3)
    gen = torch.Generator().manual_seed(0)
    views = []
    for k in range(6):
        c = syn.make_camera(W, H, R=syn.random_rotation(gen, 0.2), t=0.2 * torch.randn(3, generator=gen))
        s = dict(scene); s.update(c); views.append(ts.make_settings(s, dev, 3))
    gts = [syn.target_image(W, H, seed=10 + k).to(dev) * 0.5 + 0.25 for k in range(6)]
    for mode in ("one view", "six views", "six views, no hand-over"):
        params = ts.GaussianParams(scene, dev)
        def step(it):
            v = 0 if mode == "one view" else it % 6
            nv = 0 if mode == "one view" else (it + 1) % 6
            ts.train_step(params, views[v], gts[v], next_settings=None if "no hand" in mode else views[nv])
        for it in range(30): step(it)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for it in range(600): step(it)
        torch.cuda.synchronize()
        print(N, mode, "%.4f ms" % (1e3 * (time.perf_counter() - t0) / 600), flush=True)
