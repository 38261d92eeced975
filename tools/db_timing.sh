#!/bin/bash
# Run ON THE GPU BOX (a scratch copy of the tree): rebuilds libgsr_hip.so there with -DGSR_DB_TIMING (cycle probes in k_chunk_scatter)
# and prints where a scatter wave's time goes.  Do not run in the working tree -- it replaces the library.
cd $GRAFT_REPO_ROOT
CS=3dgs_hierarchical_training_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DGSR_DB_TIMING -Wno-unused-result -fno-slp-vectorize \
  -Wl,-soname,libgsr_hip.so $CS/gsr_kernels.hip $CS/loss_kernels.hip $CS/optim_kernels.hip $CS/knn_kernels.hip -o $CS/libgsr_hip.so 2>&1 | grep -v warning | grep -i error
python - "$@" <<'PY'
import ctypes, importlib, sys, torch
sys.path.insert(0, ".")
L = importlib.import_module("3dgs_hierarchical_training_amd._lib")
syn = importlib.import_module("3dgs_hierarchical_training_amd.synthetic")
ts = importlib.import_module("3dgs_hierarchical_training_amd.train_step")
lib = L.load()
raw = ctypes.CDLL(L.LIB_PATH if hasattr(L, "LIB_PATH") else "3dgs_hierarchical_training_amd/csrc/libgsr_hip.so")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
sc = syn.make_scene(N, 980, 545, sh_degree=deg, seed=0)
p = ts.GaussianParams(sc, dev)
st = ts.make_settings(sc, dev, deg)
with torch.no_grad():
    for i in range(6):
        ts.render(p, st)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
raw.gsr_debug_db_timing(buf)
v = list(buf)
waves, steps = v[6], v[7]
names = ["setup (rows -> LDS)", "step head (loads issued, scan)", "owners (OR + pairs)", "wait for next loads", "place", "advance"]
print(f"N={N} waves {waves} steps {steps} ({steps / max(waves,1):.1f} per wave); s_memtime ticks (100 MHz: 10 ns)")
for k, n in enumerate(names):
    per = v[k] / max(waves if k == 0 else steps, 1)
    print(f"  {n:34s} total {v[k]:12d}  per {'wave' if k == 0 else 'step'} {per:8.1f} ticks = {per * 0.01:6.2f} us")
PY
