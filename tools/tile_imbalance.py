"""Per-wave work distribution of the forward blend (one wave per 8x8 sub-tile): instances staged per sub-tile wave.
gpurun -- 'python tools/tile_imbalance.py [--clustered]'"""
import sys, importlib, torch, numpy as np
sys.path.insert(0, '.')
syn = importlib.import_module("3dgs_hierarchical_training_amd.synthetic")
ts = importlib.import_module("3dgs_hierarchical_training_amd.train_step")
R = importlib.import_module("3dgs_hierarchical_training_amd.rasterizer")
L = importlib.import_module("3dgs_hierarchical_training_amd._lib")
lib = L.load()
dev = torch.device("cuda:0")
W, H = 980, 545
sc = syn.make_scene(1_000_000, W, H, sh_degree=3, seed=0, clustered="--clustered" in sys.argv)
p = ts.GaussianParams(sc, dev); st = ts.make_settings(sc, dev, 3)
with torch.no_grad():
    ts.render(p, st)
R._sync_last()
img = R._LAST["image"]; T = 62 * 35
off = lib.gsr_image_staged_offset(W, H)
s4 = img[off:off + 16 * T].view(torch.int32).view(T, 4).cpu().numpy().astype(np.float64)
ranges, lst = R.last_binning()
n = (ranges[:, 1] - ranges[:, 0]).cpu().numpy().astype(np.float64)
w = s4.reshape(-1)
print("list length per tile: mean %.0f max %.0f" % (n.mean(), n.max()))
print("staged per sub-tile wave: mean %.0f std %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f  (max/mean %.2f)" % (
    w.mean(), w.std(), np.percentile(w, 50), np.percentile(w, 90), np.percentile(w, 99), w.max(), w.max() / w.mean()))
print("histogram (bins of 128):", np.histogram(w, bins=np.arange(0, w.max() + 129, 128))[0].tolist())
