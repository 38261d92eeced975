"""The `posed_frames_identity_camera` leg of bench.py on its own (debugging aid): FRAMES / STEPS from the environment."""
import importlib, sys, json, os, torch
sys.path.insert(0, ".")
import bench
syn = importlib.import_module("3dgs_hierarchical_training_amd.synthetic")
ts = importlib.import_module("3dgs_hierarchical_training_amd.train_step")
L = importlib.import_module("3dgs_hierarchical_training_amd._lib")
lib = L.load()
dev = torch.device("cuda:0")
sc = syn.make_scene(1_000_000, 980, 545, sh_degree=3, seed=0)
st = ts.make_settings(sc, dev, 3)
out = bench.posed_frames_leg(ts, lib, sc, st, dev, steps=int(os.environ.get("STEPS", "192")), warmup=32, frames=int(os.environ.get("FRAMES", "8")))
print(os.environ.get("TAG", ""), {k: (round(v["ms_per_step"], 3), round(v["blend_fwd_us"], 1), v["view_cache_hit_rate"], v.get("pose_optimizer")) for k, v in out.items() if isinstance(v, dict)})
