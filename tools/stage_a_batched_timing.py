"""Stage A at the reference's counts (39 pairs x (1000 + 300) iterations, 130 k-Gaussian single-image models @980x545) on one GPU:
batched launch chains (GsrBatch) against round 2's two-streams mode, plus per-iteration times by batch size."""
import importlib, json, sys, time, torch
sys.path.insert(0, '.')
sequence = importlib.import_module("3dgs_hierarchical_training_amd.sequence")
stage_a = importlib.import_module("3dgs_hierarchical_training_amd.stage_a")
rs = importlib.import_module("3dgs_hierarchical_training_amd.run_segments")
host = importlib.import_module("3dgs_hierarchical_training_amd.host"); host.cap_host_threads()
dev = torch.device("cuda:0")
import os
_L = importlib.import_module("3dgs_hierarchical_training_amd._lib")
for _kv in os.environ.get("GSR_OPTS", "").split(","):     # A/B runs: GSR_OPTS="tile_sort=0"
    if "=" in _kv:
        _k, _v = _kv.split("="); assert _L.load().gsr_set_option(_k.encode(), int(_v)) == 0
ONLY_DEFAULT = os.environ.get("STAGE_A_ONLY_DEFAULT") == "1"
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
img_it, pose_it = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1000, 300)
seq = sequence.FrameSequence(frames, 400000, 980, 545, dev, seed=0)
for f in range(frames):
    seq.target(f); seq.depth(f)
out = {}
# per-iteration cost by batch size (no early exit: fixed 200 + 100 iterations)
for B in (() if ONLY_DEFAULT else (1, 2, 4, 8)):
    pairs = list(range(B))
    stage_a.fit_pairs_batched(seq, pairs, dev, n_points=130000, single_image_iters=20, pose_iters=20)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    stage_a.fit_pairs_batched(seq, pairs, dev, n_points=130000, single_image_iters=200, pose_iters=0)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    stage_a.fit_pairs_batched(seq, pairs, dev, n_points=130000, single_image_iters=200, pose_iters=100)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    out[f"B={B}"] = {"image_iteration_ms_per_pair": 1e3 * (t1 - t0) / 200 / B, "pose_iteration_ms_per_pair": 1e3 * ((t2 - t1) - (t1 - t0)) / 100 / B}
    print(B, out[f"B={B}"], flush=True)
for mode, batch, conc in ((("default (auto)", 0, 2),) if ONLY_DEFAULT else (("batched x8", 8, 1), ("batched x4, two chains at a time", 4, 2), ("batched x8, two chains at a time", 8, 2),
                          ("batched x4", 4, 1), ("two streams (round 2)", 1, 2), ("default (auto)", 0, 2))):
    cfg = rs.HTConfig(frames=frames, stage_a_batch=batch, stage_a_concurrency=conc)
    seq.pose_table = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rec = rs.run_stage_a_on(seq, cfg, dev, (130000, img_it, pose_it), 0, 1, log=lambda r: None)
    torch.cuda.synchronize()
    out[mode] = {"seconds": time.perf_counter() - t0, "pairs": frames - 1, "max_abs_pose_error": rec["max_abs_pose_error"], "identity_guess_error": rec["identity_guess_error"]}
    print(mode, out[mode], flush=True)
print("JSON", json.dumps(out))
