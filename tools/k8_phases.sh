#!/bin/bash
# Run ON THE GPU BOX (a scratch copy of the tree): rebuilds libgsr_hip.so there with -DGSR_K8_PHASES (cycle probes in k_blend_bwd2) and
# prints where a wave of the backward blend spends its time.  Do not run in the working tree -- it replaces the library.
#   gpurun -- 'bash tools/k8_phases.sh [N] [sh degree]'
cd $GRAFT_REPO_ROOT
CS=3dgs_hierarchical_training_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DGSR_K8_PHASES -Wno-unused-result -fno-slp-vectorize \
  -Wl,-soname,libgsr_hip.so $CS/gsr_kernels.hip $CS/loss_kernels.hip $CS/optim_kernels.hip $CS/knn_kernels.hip -o $CS/libgsr_hip.so 2>&1 | grep -v warning | grep -i error
python - "$@" <<'PY'
import ctypes, importlib, sys, time, torch
import numpy as np
sys.path.insert(0, ".")
L = importlib.import_module("3dgs_hierarchical_training_amd._lib")
syn = importlib.import_module("3dgs_hierarchical_training_amd.synthetic")
ts = importlib.import_module("3dgs_hierarchical_training_amd.train_step")
host = importlib.import_module("3dgs_hierarchical_training_amd.host"); host.cap_host_threads()
lib = L.load()
raw = ctypes.CDLL("3dgs_hierarchical_training_amd/csrc/libgsr_hip.so")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
W, H = 980, 545
sc = syn.make_scene(N, W, H, sh_degree=deg, seed=0)
p = ts.GaussianParams(sc, dev)
st = ts.make_settings(sc, dev, deg)
gt = syn.target_image(W, H, seed=10).to(dev) * 0.5 + 0.25
for i in range(12):
    ts.train_step(p, st, gt, next_settings=st)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 8192))()
raw.gsr_debug_k8_phases(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.float64)
a = a[a[:, 7] > 0]
names = ["pull / list bookkeeping", "pixel state in + barriers", "staging (record wait, box tests, LDS, barriers)", "visits", "barrier behind the visits", "flush"]
life = a[:, 7] - a[:, 6]
tot = a[:, :6].sum()
print(f"N={N} degree {deg}: {len(a)} waves of the last launch; cycle-counter ticks; wave lifetime mean {life.mean():.0f} p10 {np.percentile(life, 10):.0f} p90 {np.percentile(life, 90):.0f}")
for k, n in enumerate(names):
    print(f"  {n:52s} mean {a[:, k].mean():9.0f}  ({100 * a[:, k].sum() / tot:4.1f} %)")
print(f"  accounted {100 * tot / life.sum():.1f} % of the lifetimes (the rest: exit polling of the empty lists)")
PY
