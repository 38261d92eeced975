#!/bin/bash
# Run ON THE GPU BOX (a scratch copy of the tree): rebuilds libgsr_hip.so there with -DGSR_OS_TIMING (cycle probes in k_onesweep) and
# prints where a workgroup of the depth sort's LAST pass spends its time.  Do not run in the working tree -- it replaces the library.
cd $GRAFT_REPO_ROOT
CS=3dgs_hierarchical_training_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DGSR_OS_TIMING $2 -Wno-unused-result -fno-slp-vectorize \
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
dev = torch.device("cuda:0")
sc = syn.make_scene(N, 980, 545, sh_degree=3, seed=0)
p = ts.GaussianParams(sc, dev)
st = ts.make_settings(sc, dev, 3)
with torch.no_grad():
    for i in range(6):
        ts.render(p, st)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 4096))()
raw.gsr_debug_os_timing(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.float64)
nb = (N + 4095) // 4096
a = a[:nb]
names = ["issue the loads", "loads landed + ranking + barrier", "publish + two block scans", "look-back + barrier", "tile to LDS in digit order", "global stores issued"]
life = a[:, 7] - a[:, 6]
t0 = a[:, 6].min()
print(f"N={N}: {nb} workgroups of the last u32 pass; cycle-counter ticks; first start to last end {a[:, 7].max() - t0:.0f}")
print(f"  workgroup lifetime mean {life.mean():.0f} p10 {np.percentile(life, 10):.0f} p50 {np.percentile(life, 50):.0f} p90 {np.percentile(life, 90):.0f} max {life.max():.0f}")
print(f"  start offsets: p50 {np.percentile(a[:, 6] - t0, 50):.0f} p90 {np.percentile(a[:, 6] - t0, 90):.0f} max {(a[:, 6] - t0).max():.0f}")
for k, n in enumerate(names):
    print(f"  {n:40s} mean {a[:, k].mean():8.0f}  p90 {np.percentile(a[:, k], 90):8.0f}  max {a[:, k].max():8.0f}")
gb = (ctypes.c_ulonglong * (8 * 512))()
raw.gsr_debug_gh_timing(gb)
g = np.frombuffer(gb, dtype=np.uint64).reshape(512, 8).astype(np.float64)
g = g[g[:, 7] > 0]
gl = g[:, 7] - g[:, 6]
print(f"histogram kernel: {len(g)} workgroups; lifetime mean {gl.mean():.0f} p10 {np.percentile(gl, 10):.0f} p90 {np.percentile(gl, 90):.0f} max {gl.max():.0f}")
for k, n in enumerate(["rider (last workgroup)", "status clear issued + tables zeroed", "keys counted into LDS", "barrier", "global adds issued", "global adds acknowledged"]):
    print(f"  {n:40s} mean {g[:, k].mean():8.0f}  p90 {np.percentile(g[:, k], 90):8.0f}  max {g[:, k].max():8.0f}")
q = np.argsort(a[:, 7])[-5:]
print("  last five to finish: workgroup, start, end, look-back ticks:", [(int(i), int(a[i, 6] - t0), int(a[i, 7] - t0), int(a[i, 3])) for i in q])
PY
