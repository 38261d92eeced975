"""Per-visit latency of the blend kernels when a wave has its SIMD to itself: ONE 16x16 tile, n faint wide splats that every
pixel takes and none saturates -- the four waves of the forward blend (two of the backward) walk n instances each on an otherwise
empty chip.  Prints kernel time / n for the forward and the backward blend (library HIP-event profile)."""
import ctypes as C
import importlib
import sys

import torch

sys.path.insert(0, ".")
ts = importlib.import_module("3dgs_hierarchical_training_amd.train_step")
syn = importlib.import_module("3dgs_hierarchical_training_amd.synthetic")
L = importlib.import_module("3dgs_hierarchical_training_amd._lib")
lib = L.load()
dev = torch.device("cuda:0")
import os
CASES = [(16, 512), (16, 1024), (256, 512), (512, 512), (736, 512), (1024, 512)] if os.environ.get("GSR_UNIFORM") else [(16, 512), (16, 1024), (16, 1536)]
for W, n in CASES:
    H = W
    sc = syn.make_scene(n, W, H, sh_degree=0, seed=0)
    cam_fx = 0.5 * W / sc["tanfovx"]
    z = torch.linspace(2.0, 4.0, n)
    sc["means3D"] = torch.stack((torch.zeros(n), torch.zeros(n), z), 1).contiguous()
    sc["scales"] = (1.25 * W * z / cam_fx)[:, None].repeat(1, 3).contiguous()     # sigma = 1.25 W px: G >= 0.85 over the whole image
    sc["opacities"] = torch.full((n, 1), 0.005)                                    # above 1/255 everywhere, exp(-0.005 n) stays above 1e-4
    p = ts.GaussianParams(sc, dev)
    st = ts.make_settings(sc, dev, 0)
    gt = syn.target_image(W, H, seed=1).to(dev)
    for _ in range(5):
        ts.train_step(p, st, gt)
    lib.gsr_set_option(b"profile", 1)
    out = {}
    for name in ("blend_fwd", "blend_bwd"):
        tot, cnt = C.c_double(0), C.c_int64(0)
        lib.gsr_profile_read(name.encode(), C.byref(tot), C.byref(cnt))
    for _ in range(20):
        ts.train_step(p, st, gt)
    torch.cuda.synchronize()
    for name in ("blend_fwd", "blend_bwd"):
        tot, cnt = C.c_double(0), C.c_int64(0)
        lib.gsr_profile_read(name.encode(), C.byref(tot), C.byref(cnt))
        out[name] = 1e3 * tot.value / max(1, cnt.value)
    lib.gsr_set_option(b"profile", 0)
    waves = 4 * ((W + 15) // 16) ** 2
    print(f"{W}x{H} ({waves} forward-blend waves = {waves / 1024:.2f} per SIMD), n {n}: forward blend {out['blend_fwd']:.1f} us = {1e3 * out['blend_fwd'] / n:.0f} ns per instance; backward blend {out['blend_bwd']:.1f} us = "
          f"{1e3 * out['blend_bwd'] / n:.0f} ns per instance")
