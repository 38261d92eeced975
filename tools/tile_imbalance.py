import sys, importlib, torch, numpy as np
sys.path.insert(0,'.')
syn = importlib.import_module("3dgs_hierarchical_training_amd.synthetic")
ts = importlib.import_module("3dgs_hierarchical_training_amd.train_step")
R = importlib.import_module("3dgs_hierarchical_training_amd.rasterizer")
L = importlib.import_module("3dgs_hierarchical_training_amd._lib")
lib = L.load()
dev = torch.device("cuda:0")
sc = syn.make_scene(1_000_000, 980, 545, sh_degree=3, seed=0)
p = ts.GaussianParams(sc, dev); st = ts.make_settings(sc, dev, 3)
with torch.no_grad(): ts.render(p, st)
img = R._LAST["image"]; W,H=980,545; T=62*35
off = lib.gsr_image_staged_offset(W,H)
staged = img[off:off+4*T].view(torch.int32).cpu().numpy().astype(np.float64)
print("staged per tile: mean %.0f std %.0f min %.0f max %.0f cv %.3f" % (staged.mean(), staged.std(), staged.min(), staged.max(), staged.std()/staged.mean()))
# static mapping: block b -> tile xcd_tile(b); CU assignment ~ round robin over 256 CUs in dispatch order: b -> XCD b%8, CU within XCD (b//8)%32
per = (T+7)//8
load = np.zeros(256)
for b in range(8*per):
    t = (b & 7)*per + (b >> 3)
    if t < T: load[(b % 8)*32 + ((b//8) % 32)] += staged[t]
print("per-CU load: mean %.0f max %.0f  max/mean %.3f" % (load.mean(), load.max(), load.max()/load.mean()))
