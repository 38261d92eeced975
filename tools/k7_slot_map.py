"""Where the hardware puts the forward blend's workgroups, and how predictable a wave's work is (experiment build only).

Needs the library compiled with -DGSR_K6_TIMING (tools/k7_slot_map.sh).  For several consecutive training steps on the bench scene
(8 rotating views) it records, per workgroup of k_blend_fwd_w6 (blockIdx = xcd + 8 kslot): XCC_ID, HW_ID (SE / SH / CU / SIMD / wave
slot), start / end on the 100 MHz clock, list length, staged length, visits.  -> gpurun_out/k7_slot_map.npz for offline analysis."""
import ctypes as C
import importlib
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
syn = importlib.import_module("3dgs_hierarchical_training_amd.synthetic")
ts = importlib.import_module("3dgs_hierarchical_training_amd.train_step")
L = importlib.import_module("3dgs_hierarchical_training_amd._lib")
bench = importlib.import_module("bench")

lib = L.load()
dev = torch.device("cuda:0")
N, W, H, deg = 1_000_000, 980, 545, 3
scene = syn.make_scene(N, W, H, sh_degree=deg, seed=0)
p = ts.GaussianParams(scene, dev)
views = bench.make_views(syn, ts, scene, dev, deg, 8)
raw = C.CDLL(L.LIB_PATH)
T = ((W + 15) // 16) * ((H + 15) // 16)
nb6 = min(65536, 8 * 4 * ((T + 7) // 8 + 8))
out = {}
for step in range(26):
    st, gt = views[step % 8]
    ts.train_step(p, st, gt)
    torch.cuda.synchronize()
    if step < 8:
        continue
    buf = np.zeros(4 * nb6, dtype=np.uint64)
    assert raw.gsr_debug_k6_timing(buf.ctypes.data_as(C.c_void_p), C.c_int(nb6)) == 0
    cnt = np.zeros(2 * nb6, dtype=np.uint32)
    assert raw.gsr_debug_k6_counts(cnt.ctypes.data_as(C.c_void_p), C.c_int(nb6)) == 0
    out[f"t_{step}"] = buf.reshape(nb6, 4).copy()
    out[f"c_{step}"] = cnt.reshape(nb6, 2).copy()
np.savez_compressed("gpurun_out/k7_slot_map.npz", **out)
d = out["t_8"]
hw = (d[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
xcc = ((d[:, 2] >> np.uint64(32)).astype(np.int64)) & 0xf
live = d[:, 1] > 0
print("workgroups", int(live.sum()), "xcc == blockIdx & 7 for", int((xcc[live] == (np.arange(nb6)[live] & 7)).sum()))
