#!/bin/bash
# Run ON THE GPU BOX (a scratch copy of the tree): rebuilds libgsr_hip.so there with -DGSR_K9_TIMING (cycle probes in
# k_preprocess_bwd) and prints where a wave of the per-Gaussian backward spends its time in the fused training step.
# Do not run in the working tree -- it replaces the library.        gpurun -- 'bash tools/k9_timing.sh [N] [extra -D flags]'
cd $GRAFT_REPO_ROOT
CS=3dgs_hierarchical_training_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DGSR_K9_TIMING $2 -Wno-unused-result -fno-slp-vectorize \
  -Wl,-soname,libgsr_hip.so $CS/gsr_kernels.hip $CS/loss_kernels.hip $CS/optim_kernels.hip $CS/knn_kernels.hip -o $CS/libgsr_hip.so 2>&1 | grep -v warning | grep -i error
python - "$@" <<'PY'
import ctypes, importlib, sys, time, torch
sys.path.insert(0, ".")
L = importlib.import_module("3dgs_hierarchical_training_amd._lib")
syn = importlib.import_module("3dgs_hierarchical_training_amd.synthetic")
ts = importlib.import_module("3dgs_hierarchical_training_amd.train_step")
host = importlib.import_module("3dgs_hierarchical_training_amd.host"); host.cap_host_threads()
lib = L.load()
raw = ctypes.CDLL("3dgs_hierarchical_training_amd/csrc/libgsr_hip.so")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda:0")
W, H = 980, 545
sc = syn.make_scene(N, W, H, sh_degree=3, seed=0)
p = ts.GaussianParams(sc, dev)
st = ts.make_settings(sc, dev, 3)
gt = syn.target_image(W, H, seed=10).to(dev) * 0.5 + 0.25
for i in range(10):
    ts.train_step(p, st, gt, next_settings=st)
torch.cuda.synchronize()
import numpy as np
NW = 16384
buf = (ctypes.c_ulonglong * (8 * NW))()
steps = 20
t0 = time.perf_counter()
for i in range(steps):
    ts.train_step(p, st, gt, next_settings=st)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
raw.gsr_debug_k9_timing(buf, 0)
a = np.frombuffer(buf, dtype=np.uint64).reshape(NW, 8).astype(np.float64)
waves = min(NW, (N + 63) // 64)
a = a[:waves]
names = ["SH rows in (issue + wait + barrier)", "ggrad / parameters in + derivative chain", "densification statistics + small groups' Adam",
         "barrier + row streams (f_dc, f_rest Adam)", "next-view preprocess + its stores", "digit counts + status clear"]
v = a[:, :6].sum(0)
tot = v.sum()
span = a[:, 7].max() - a[:, 6].min()
print(f"N={N}: step {dt * 1e3:.3f} ms; {waves} waves of the last launch; cycle counter ticks; first start to last end {span:.0f} ticks")
life = a[:, 7] - a[:, 6]
print(f"  wave lifetime mean {life.mean():.0f} p10 {np.percentile(life, 10):.0f} p50 {np.percentile(life, 50):.0f} p90 {np.percentile(life, 90):.0f}")
for k, n in enumerate(names):
    per = v[k] / waves
    print(f"  {n:48s} {per:8.1f} ticks ({100.0 * v[k] / tot:4.1f} %)")
PY
