"""Per-stage kernel times (library's HIP-event profile) of one stage-A image iteration: B models per launch chain vs one."""
import ctypes as C, importlib, sys, time, torch
sys.path.insert(0, '.')
sequence = importlib.import_module("3dgs_hierarchical_training_amd.sequence")
ts = importlib.import_module("3dgs_hierarchical_training_amd.train_step")
bt = importlib.import_module("3dgs_hierarchical_training_amd.batched")
L = importlib.import_module("3dgs_hierarchical_training_amd._lib")
raster = importlib.import_module("3dgs_hierarchical_training_amd.rasterizer")
host = importlib.import_module("3dgs_hierarchical_training_amd.host"); host.cap_host_threads()
lib = L.load()
import os
for kv in os.environ.get("GSR_OPTS", "").split(","):
    if "=" in kv:
        k, v = kv.split("="); print("option", k, v, lib.gsr_set_option(k.encode(), int(v)))
BS = [int(x) for x in os.environ.get("GSR_BS", "1,2,4,8").split(",")]
dev = torch.device("cuda:0")
STAGES = ["preprocess_fwd", "sort_depth", "scan", "emit", "sort_tile", "ranges", "blend_fwd", "blend_bwd", "preprocess_bwd"]
seq = sequence.FrameSequence(9, 400000, 980, 545, dev, seed=0)
def read():
    out = {}
    for n in STAGES:
        tot, cnt = C.c_double(0), C.c_int64(0)
        lib.gsr_profile_read(n.encode(), C.byref(tot), C.byref(cnt))
        out[n] = 1e3 * tot.value / max(1, cnt.value)
    return out
for B in BS:
    scenes = [seq.pixel_scene(p, stride=2, seed=0) for p in range(B)]
    ident1 = ts.with_sh_degree(seq.settings_for_pose(torch.eye(4)), 0)
    if B == 1:
        params = ts.GaussianParams(scenes[0], dev); params.active_sh_degree = 0
        ident, tgt = ident1, seq.target(0)
    else:
        params = bt.BatchedGaussianParams(scenes, dev); params.active_sh_degree = 0
        ident, tgt = bt.batch_settings([ident1] * B, dev), torch.stack([seq.target(p) for p in range(B)])
    for _ in range(30): ts.train_step(params, ident, tgt, next_settings=ident)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): ts.train_step(params, ident, tgt, next_settings=ident)
    torch.cuda.synchronize(); wall = 1e3 * (time.perf_counter() - t0) / 200
    lib.gsr_set_option(b"profile", 1); read()
    for _ in range(10): ts.train_step(params, ident, tgt, next_settings=ident)
    torch.cuda.synchronize(); lib.gsr_set_option(b"profile", 0)
    st = read()
    with torch.no_grad(): ts.render(params, ident, fused_activations=True)
    info = raster.last_call_info()
    print(f"B={B} N={params.num_points} R={info['num_rendered']} R_eff={info['staged']} wall {wall:.3f} ms/step = {wall / B:.3f} per pair; rasterizer kernels {sum(st.values()):.0f} us:",
          {k: round(v) for k, v in st.items()}, flush=True)
