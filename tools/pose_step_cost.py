"""Cost of refining the camera pose inside the train step (the reference's camera_optimizer): two frames alternating, with and
without a CameraPoseState per frame.   gpurun -- 'python tools/pose_step_cost.py [N]'"""
import importlib, sys, time, torch
sys.path.insert(0, '.')
syn = importlib.import_module('3dgs_hierarchical_training_amd.synthetic')
ts = importlib.import_module('3dgs_hierarchical_training_amd.train_step')
host = importlib.import_module("3dgs_hierarchical_training_amd.host"); host.cap_host_threads()
dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
W, H = 980, 545
scene = syn.make_scene(N, W, H, sh_degree=3, seed=0)
gen = torch.Generator().manual_seed(0)
views, w2c = [], []
for k in range(2):
    R, t = syn.random_rotation(gen, 0.05), 0.05 * torch.randn(3, generator=gen)
    c = syn.make_camera(W, H, R=R, t=t)
    s = dict(scene); s.update(c); views.append(ts.make_settings(s, dev, 3))
    M = torch.eye(4); M[:3, :3] = R; M[:3, 3] = t; w2c.append(M)
gts = [syn.target_image(W, H, seed=10 + k).to(dev) for k in range(2)]
for mode in ("fixed cameras", "poses refined"):
    params = ts.GaussianParams(scene, dev)
    states = [ts.CameraPoseState(views[k], w2c[k], dev, lr=1e-5) for k in range(2)] if mode == "poses refined" else None
    def step(it):
        v, nv = it % 2, (it + 1) % 2
        if states is None:
            ts.train_step(params, views[v], gts[v], next_settings=views[nv])
        else:
            ts.train_step(params, states[v].settings, gts[v], pose=states[v], next_pose=states[nv])
    for it in range(20): step(it)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(200): step(it)
    torch.cuda.synchronize()
    print(f"N {N}, {mode}: {1e3 * (time.perf_counter() - t0) / 200:.4f} ms per step", flush=True)
