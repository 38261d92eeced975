"""CPU: three independent restatements must agree -- C oracle (hand-derived backward), dense torch-autograd
oracle, and the kernels' binary32 arithmetic driven by the host harness -- plus finite differences and the
committed regression vectors."""
import os

import numpy as np
import pytest
import torch

import parity
from oracle import binding, torch_oracle


@pytest.mark.parametrize("posed,mode,deg", [(False, "sh", 3), (True, "sh", 2), (True, "pre", 0), (False, "mixed", 0)])
def test_c_oracle_vs_autograd_oracle(posed, mode, deg):
    N, W, H = 300, 64, 48
    sc = parity.syn.make_scene(N, W, H, sh_degree=deg, seed=3, posed=posed, sigma_px=4.0)
    kw = parity.scene_kwargs(sc, mode, bg=(0.1, 0.2, 0.3))
    o = binding.OracleRender(**kw)
    color, radii, depth, alpha = o.forward()
    gc, gd, ga = parity.upstream_grads(H, W, seed=0, depth_scale=1.0, alpha_scale=1.0)
    g = o.backward(gc, gd, ga)
    inp = {k: (v.numpy() if hasattr(v, "numpy") else v) for k, v in kw.items()}
    t = torch_oracle.render_from_f32(inp, (gc, gd, ga))
    assert np.abs(color - t["color"]).max() < 1e-6 and np.abs(alpha - t["alpha"]).max() < 1e-6
    assert np.abs(depth - t["depth"]).max() < 1e-5 and np.array_equal(radii, t["radii"])
    for k, ref in t["grads"].items():
        got = g[k].reshape(ref.shape)
        assert np.abs(got - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), k


def test_finite_differences_tiny():
    """Central differences of the float64 torch oracle's own forward on a 12-Gaussian scene."""
    N, W, H = 12, 32, 32
    sc = parity.syn.make_scene(N, W, H, sh_degree=1, seed=9, posed=True, sigma_px=5.0, frac_behind=0.0)
    kw = parity.scene_kwargs(sc, "sh", bg=(0.2, 0.2, 0.2))
    inp = {k: (v.numpy().astype(np.float64) if hasattr(v, "numpy") else v) for k, v in kw.items()}
    rng = np.random.default_rng(1)
    gc = rng.standard_normal((3, H, W))
    base = torch_oracle.render_from_f32(inp, (gc, None, None))

    def loss(mod):
        return float((torch_oracle.render_from_f32(mod)["color"] * gc).sum())

    for key, idx in [("means3D", (3, 0)), ("scales", (5, 1)), ("rotations", (2, 2)), ("opacities", (7, 0))]:
        eps = 1e-5
        p, m = dict(inp), dict(inp)
        p[key] = inp[key].copy(); m[key] = inp[key].copy()
        p[key][idx] += eps; m[key][idx] -= eps
        fd = (loss(p) - loss(m)) / (2 * eps)
        an = base["grads"][key][idx]
        assert abs(fd - an) <= 2e-3 * max(1.0, abs(an)), (key, fd, an)


@pytest.mark.parametrize("N,W,H,deg,posed,mode", [(20000, 320, 240, 3, True, "sh"), (8000, 200, 150, 1, False, "pre"),
                                                   (40000, 980, 545, 3, True, "sh")])
def test_kernel_arithmetic_on_host_vs_oracle(N, W, H, deg, posed, mode):
    """csrc/gsr_math.h (what the HIP kernels execute) driven sequentially on the host, at the parity
    tolerances of BASELINE.json: forward 1e-5 abs, gradients 1e-4 rel."""
    sc = parity.syn.make_scene(N, W, H, sh_degree=deg, seed=1, posed=posed)
    kw = parity.scene_kwargs(sc, mode, bg=(0.2, 0.1, 0.3))
    o = binding.OracleRender(**kw)
    o.forward()
    n_amb = int((o.px_ambig != 0).sum())
    emu = parity.hostemu_run(o)
    rep = parity.check_forward(emu["fwd"], o, "hostemu")     # adopts, per rounding-edge pixel, the branch the kernels took
    print({k: v for k, v in rep.items() if k != "grad_mask"}, "edge pixels", n_amb)
    gc, gd, ga = parity.upstream_grads(H, W)
    keep = rep["grad_mask"]                                   # everything but the (few) unresolved pixels
    gc *= keep[None]; gd *= keep; ga *= keep
    ref = o.backward(gc, gd, ga)
    emu = parity.hostemu_run(o, (gc, gd, ga))
    got = {k: v for k, v in emu["grads"].items() if kw.get(k) is not None or k in ("means2D", "opacities", "means3D")}
    got["viewmatrix"], got["projmatrix"] = emu["grads"]["viewmatrix"], emu["grads"]["projmatrix"]
    if mode == "sh":
        got["campos"] = emu["grads"]["campos"]
    parity.check_grads(got, ref, "hostemu", elem_bad_max=0.0)     # fixed accumulation order: no entry may be off


def test_oracle_regression_vs_committed_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "oracle_small_deg3.npz"))
    N, W, H, deg = int(g["N"]), int(g["W"]), int(g["H"]), int(g["deg"])
    sc = parity.syn.make_scene(N, W, H, sh_degree=deg, seed=int(g["seed"]), posed=bool(g["posed"]))
    kw = parity.scene_kwargs(sc, "sh")
    o = binding.OracleRender(**kw)
    color, radii, depth, alpha = o.forward()
    assert o.num_rendered == int(g["num_rendered"])
    assert np.array_equal(radii, g["radii"].astype(np.int32))
    assert np.abs(color - g["color"]).max() == 0 and np.abs(depth - g["depth"]).max() == 0      # float32 fixture, same code
    assert abs(float(color.astype(np.float64).sum()) - float(g["color_sum"])) < 1e-6 * abs(float(g["color_sum"]))


def test_degenerate_inputs_oracle():
    sc = parity.syn.make_scene(50, 48, 32, sh_degree=0, seed=2)
    kw = parity.scene_kwargs(sc, "sh")
    kw["means3D"] = kw["means3D"].clone(); kw["means3D"][:, 2] = -3.0
    o = binding.OracleRender(**kw)
    color, radii, depth, alpha = o.forward()
    assert o.num_rendered == 0 and np.all(radii == 0) and np.all(color == 0)
    g = o.backward(*parity.upstream_grads(32, 48))
    assert all(np.all(v == 0) for v in g.values())


def test_tight_candidate_rect_never_loses_an_accepted_tile():
    """tile_rect_tight (the bounding box of the contribution ellipse, intersected with the reference's 3-sigma rect) is
    only an optimisation of the candidate enumeration: over 200 000 random splats -- needles up to 55:1, opacities down
    to the 1/255 threshold, sigmas from 0.4 to 400 px, centres partly off-screen -- no tile that passes the exact test
    may fall outside it."""
    import ctypes as C
    lib = parity.hostemu_lib()
    lib.hostemu_check_tight_rect.restype = C.c_longlong
    lib.hostemu_check_tight_rect.argtypes = [C.c_int, C.c_uint, C.c_int, C.c_int, C.POINTER(C.c_longlong)]
    for W, H, seed in ((980, 545, 1), (1920, 1080, 2), (256, 256, 3)):
        seen = C.c_longlong(0)
        bad = lib.hostemu_check_tight_rect(70000, seed, W, H, C.byref(seen))
        assert bad == 0 and seen.value > 100000, (W, H, bad, seen.value)


def test_exact_box_culling_is_conservative():
    """box_accept -- the test behind the image-preserving tile culling and the backward's reach bits -- may only reject a
    box in which no pixel centre receives a contribution by the blend's own rule.  600 000 random (splat, box) pairs placed
    around the splat's 1/255 contour, needles and threshold opacities included."""
    import ctypes as C
    lib = parity.hostemu_lib()
    lib.hostemu_check_box_accept.restype = C.c_longlong
    lib.hostemu_check_box_accept.argtypes = [C.c_int, C.c_uint, C.POINTER(C.c_longlong)]
    for seed in (11, 12, 13):
        seen = C.c_longlong(0)
        bad = lib.hostemu_check_box_accept(200000, seed, C.byref(seen))
        assert bad == 0 and seen.value > 50000, (seed, bad, seen.value)
