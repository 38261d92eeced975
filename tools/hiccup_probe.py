"""Which train steps stall the launching thread, and do they coincide with fresh device allocations of the caching allocator?
gpurun -- 'python tools/hiccup_probe.py [N] [steps]'"""
import importlib, sys, time
import torch
sys.path.insert(0, '.')
syn = importlib.import_module("3dgs_hierarchical_training_amd.synthetic")
ts = importlib.import_module("3dgs_hierarchical_training_amd.train_step")
R_ = importlib.import_module("3dgs_hierarchical_training_amd.rasterizer")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
import os as _os
if _os.environ.get("GSR_POLL") is not None:   # A/B of gsr_forward's busy-wait bound
    L_ = importlib.import_module("3dgs_hierarchical_training_amd._lib")
    assert L_.load().gsr_set_option(b"poll_iters", int(_os.environ["GSR_POLL"])) == 0
W, H = 980, 545
sc = syn.make_scene(N, W, H, sh_degree=3, seed=0)
params = ts.GaussianParams(sc, dev)
st = ts.make_settings(sc, dev, 3)
gt = syn.target_image(W, H, seed=1).to(dev)
for _ in range(5):
    ts.train_step(params, st, gt, next_settings=st)
torch.cuda.synchronize()
import gc, os
if os.environ.get("NOGC"):
    gc.collect(); gc.disable()
gcs = []
def _cb(phase, info):
    if phase == "start":
        gcs.append((len(stamps), info["generation"]))
gc.callbacks.append(_cb)
stamps, allocs, caps = [], [], []
for i in range(steps):
    stamps.append(time.perf_counter())
    ts.train_step(params, st, gt, next_settings=st)
    ms = torch.cuda.memory_stats(dev)
    allocs.append((ms["num_device_alloc"], ms["num_device_free"], ms["reserved_bytes.all.current"] >> 20))
    caps.append(R_._LAST.get("binning_capacity", None) if hasattr(R_, "_LAST") else None)
torch.cuda.synchronize()
stamps.append(time.perf_counter())
d = [1e3 * (b - a) for a, b in zip(stamps[:-1], stamps[1:])]
med = sorted(d)[len(d) // 2]
print(f"N {N}: median {med:.3f} ms, mean {sum(d) / len(d):.3f} ms, max {max(d):.3f} ms; device allocs {allocs[0][0]} -> {allocs[-1][0]}, reserved {allocs[0][2]} -> {allocs[-1][2]} MiB")
slow = [(i, x) for i, x in enumerate(d) if x > 1.2 * med]
print(f"steps above 1.2 x median: {len(slow)} of {len(d)}, excess {sum(x - med for _, x in slow):.2f} ms of {sum(d):.1f} ms total; their indices: {[i for i, _ in slow][:60]}")
if os.environ.get("QUIET"):
    sys.exit(0)
print("gc runs (step, generation):", gcs[:40])
for i, x in enumerate(d):
    if x > 1.5 * med:
        print(f"  step {i}: {x:.3f} ms  allocs {allocs[i - 1][0] if i else '-'} -> {allocs[i][0]}  frees -> {allocs[i][1]}  reserved {allocs[i][2]} MiB cap {caps[i]}")
