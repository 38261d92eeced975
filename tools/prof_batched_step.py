import importlib, sys, torch
sys.path.insert(0, ".")
sequence = importlib.import_module("3dgs_hierarchical_training_amd.sequence")
bt = importlib.import_module("3dgs_hierarchical_training_amd.batched")
ts = importlib.import_module("3dgs_hierarchical_training_amd.train_step")
host = importlib.import_module("3dgs_hierarchical_training_amd.host"); host.cap_host_threads()
dev = torch.device("cuda:0")
import os
L = importlib.import_module("3dgs_hierarchical_training_amd._lib")
for kv in os.environ.get("GSR_OPTS", "").split(","):
    if "=" in kv:
        k, v = kv.split("="); assert L.load().gsr_set_option(k.encode(), int(v)) == 0
seq = sequence.FrameSequence(12, 400000, 980, 545, dev, seed=0)
B = 8
pairs = list(range(B))
stride = max(1, int(round((seq.W * seq.H / 130000) ** 0.5)))
scenes = [seq.pixel_scene(p, stride=stride, seed=0) for p in pairs]
params = bt.BatchedGaussianParams(scenes, dev)
params.active_sh_degree = 0
ident1 = ts.with_sh_degree(seq.settings_for_pose(torch.eye(4)), 0)
ident = bt.batch_settings([ident1] * B, dev)
tgt0 = torch.stack([seq.target(p) for p in pairs])
for it in range(10):
    ts.train_step(params, ident, tgt0, next_settings=ident)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for it in range(60):
    ts.train_step(params, ident, tgt0, next_settings=ident)
torch.cuda.synchronize()
print("GSR_OPTS", os.environ.get("GSR_OPTS", ""), "batched step ms", 1e3 * (time.perf_counter() - t0) / 60, flush=True)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for it in range(5):
        ts.train_step(params, ident, tgt0, next_settings=ident)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
for e in prof.events():
    if "copy" in e.name.lower() or "Memcpy" in e.name:
        print(e.name, e.device_type, getattr(e, "input_shapes", None), [s for s in (e.stack or [])][:6])
        break
print(prof.key_averages(group_by_stack_n=6).table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=50, max_src_column_width=110))
