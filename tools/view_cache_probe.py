#!/usr/bin/env python3
"""Forward-blend time with the balanced placement off / keyed by the caller's frame id / keyed by pose, on the reference's calling
convention (identity camera, the frame's pose as points_transform, drifting a little after every render): 1 M Gaussians @980x545, 8
frames drawn at random, no training (the model is static, so the previous visit of a frame predicts this one almost exactly)."""
import ctypes as C
import importlib
import random
import sys

import torch

sys.path.insert(0, ".")
syn = importlib.import_module("3dgs_hierarchical_training_amd.synthetic")
ts = importlib.import_module("3dgs_hierarchical_training_amd.train_step")
L = importlib.import_module("3dgs_hierarchical_training_amd._lib")
R = importlib.import_module("3dgs_hierarchical_training_amd.rasterizer")
lib = L.load()
dev = torch.device("cuda:0")
N, W, H, F = 1_000_000, 980, 545, 8
sc = syn.make_scene(N, W, H, sh_degree=3, seed=0)
p = ts.GaussianParams(sc, dev)
st = ts.make_settings(sc, dev, 3)
gen = torch.Generator().manual_seed(5)
base = []
for f in range(F):
    M = torch.eye(4)
    if f:
        M[:3, :3] = syn.random_rotation(gen, 0.03)
        M[:3, 3] = 0.03 * torch.randn(3, generator=gen)
    base.append(M[:3].contiguous())
z = torch.zeros(N, 3, device=dev)


def read(name):
    tot, cnt = C.c_double(0), C.c_int64(0)
    lib.gsr_profile_read(name.encode(), C.byref(tot), C.byref(cnt))
    return tot.value, cnt.value


def stats():
    out = (C.c_int64 * 4)()
    lib.gsr_debug_view_cache_stats(W, H, out)
    return out[0], out[1], out[2]


def run(balance, mode, steps=160, drift=1e-4, idle_ms=0.0):
    lib.gsr_set_option(b"blend_balance", balance)
    rng = random.Random(3)
    poses = [b.clone() for b in base]
    g2 = torch.Generator().manual_seed(9)
    with torch.no_grad():
        for i in range(steps + 32):
            if i == 32:
                lib.gsr_set_option(b"profile", 1)
                read("blend_fwd")
                s0 = stats()
            f = rng.randrange(F)
            if idle_ms:
                torch.cuda.synchronize()
                import time as _t
                _t.sleep(idle_ms * 1e-3)
            poses[f] = poses[f] + drift * torch.randn(3, 4, generator=g2)
            R.rasterize_gaussians_raw(p._xyz, z, p._features_dc, p._features_rest, p._opacity, p._scaling, p._rotation, st,
                                      points_transform=poses[f].to(dev), view_id=(f + 1 + 1000 * (mode == "uid")) if mode == "uid" else 0)
    torch.cuda.synchronize()
    lib.gsr_set_option(b"profile", 0)
    tot, cnt = read("blend_fwd")
    s1 = stats()
    look = s1[0] - s0[0]
    return 1e3 * tot / max(cnt, 1), (s1[1] - s0[1]) / look if look else None, s1[2]


for rep in range(2):
    for bal, mode in ((0, "off"), (1, "uid"), (1, "pose"), (1, "pose-drift-1e-3")):
        us, hit, entries = run(bal, "pose" if mode.startswith("pose") else mode, drift=1e-3 if mode.endswith("1e-3") else 1e-4)
        print(f"{mode:16s} blend_fwd {us:7.1f} us   hit rate {hit}   entries in use {entries}")
for idle in (0.5, 2.0):
    for bal, mode in ((0, "off"), (1, "uid")):
        us, hit, entries = run(bal, mode, idle_ms=idle)
        print(f"{mode:5s} with the device idle {idle} ms between renders: blend_fwd {us:7.1f} us   hit rate {hit}")
lib.gsr_set_option(b"blend_balance", 1)

# ---- the same with TRAINING (the library's own train_step: in-kernel Adam, one PoseState per frame stepped after every render) ----------
gts = [syn.target_image(W, H, seed=60 + f).to(dev) for f in range(F)]


def run_train(balance, use_id, steps=192):
    lib.gsr_set_option(b"blend_balance", balance)
    pp = ts.GaussianParams(sc, dev)
    full = [torch.cat([b, torch.tensor([[0.0, 0.0, 0.0, 1.0]])], 0) for b in base]
    poses = [ts.PoseState(full[f], dev, lr=1e-4) for f in range(F)]
    rng = random.Random(3)
    for i in range(steps + 32):
        if i == 32:
            lib.gsr_set_option(b"profile", 1)
            read("blend_fwd")
            s0 = stats()
        f = rng.randrange(F)
        ts.train_step(pp, st, gts[f], pose=poses[f], view_id=(f + 1 + 5000) if use_id else 0, iteration=i + 1)
    torch.cuda.synchronize()
    lib.gsr_set_option(b"profile", 0)
    tot, cnt = read("blend_fwd")
    s1 = stats()
    look = s1[0] - s0[0]
    return 1e3 * tot / max(cnt, 1), (s1[1] - s0[1]) / look if look else None


for rep in range(2):
    for bal, use_id, name in ((0, False, "off"), (1, True, "uid"), (1, False, "pose")):
        us, hit = run_train(bal, use_id)
        print(f"training, {name:5s}: blend_fwd {us:7.1f} us   hit rate {hit}")
lib.gsr_set_option(b"blend_balance", 1)
