"""Sub-pixel splats and binary32: what the kernels' formulation costs against the float64 oracle, next to the public CUDA module's.

VERDICT r3 item 6b asked whether the 2.5e-5 forward allowance of the sub-pixel-splat fuzz cases was a property of binary32 or of THIS
build's formulation; VERDICT r4 item 8 asked to retire it.  Round 5 did: the kernels project in float64, round the pixel-space mean
once -- and now keep the sub-ulp remainder of that rounding (16 bits per coordinate, gsr_math.h pixel_lo_pack) and form their offsets
from tile-relative coordinates (pixel_rel).  Here the public module's formulation (binary32 throughout, d = mean - pixel from ABSOLUTE
pixel coordinates) is stated in plain numpy float32 and pushed through the same sequential host blend (tests/hostemu with absolute
pixel coordinates), next to the kernels' own arithmetic (tests/hostemu as is) and to the kernels' arithmetic of rounds 1-4 (the
remainder switched off), on the ten fuzz cases that carried the allowance.  Result: with the remainder nine of the ten are within
5e-7 of the oracle (3e-6 without it, up to 3.8e-5 in the public formulation); the tenth -- case 13, 257 sub-pixel splats piled onto a
few pixels by a 166-degree field of view -- stays at 1.3e-5 in all three: a hundred-odd non-saturating layers per pixel, binary32's
accumulation (T *= 1 - alpha, C += c alpha T), not the coordinates.  It keeps a stated allowance (tests/test_gpu_parity.py)."""
import ctypes as C

import numpy as np
import pytest
import torch

import parity
from oracle import binding

f32 = np.float32


def _public_projection_f32(kw):
    """xy (absolute pixel coordinates), conic -- binary32 statement of the published preprocess: transformPoint4x4 + 1/(w + 1e-7),
    computeCov3D (S R)^T (S R), computeCov2D with the 1.3 tan(fov) clamp and the +0.3 low-pass, conic = cov^-1, ndc2Pix."""
    P = kw["means3D"].numpy().astype(f32)
    vm, pm = kw["viewmatrix"].numpy().astype(f32), kw["projmatrix"].numpy().astype(f32)
    W, H = int(kw["image_width"]), int(kw["image_height"])
    tfx, tfy = f32(kw["tanfovx"]), f32(kw["tanfovy"])
    x, y, z = P[:, 0], P[:, 1], P[:, 2]
    tp = lambda m, c: m[0, c] * x + m[1, c] * y + m[2, c] * z + m[3, c]          # row vector times matrix, left to right
    hx, hy, hw = tp(pm, 0), tp(pm, 1), tp(pm, 3)
    pw = f32(1.0) / (hw + f32(1e-7))
    px, py = hx * pw, hy * pw
    xy = np.stack([((px + f32(1.0)) * f32(W) - f32(1.0)) * f32(0.5), ((py + f32(1.0)) * f32(H) - f32(1.0)) * f32(0.5)], 1).astype(f32)
    if kw.get("cov3D_precomp") is not None:
        c = kw["cov3D_precomp"].numpy().astype(f32)
        S3 = np.zeros((len(P), 3, 3), f32)
        S3[:, 0, 0], S3[:, 0, 1], S3[:, 0, 2], S3[:, 1, 1], S3[:, 1, 2], S3[:, 2, 2] = c.T
        S3[:, 1, 0], S3[:, 2, 0], S3[:, 2, 1] = c[:, 1], c[:, 2], c[:, 4]
    else:
        s = kw["scales"].numpy().astype(f32) * f32(kw.get("scale_modifier", 1.0))
        q = kw["rotations"].numpy().astype(f32)
        q = q / np.sqrt((q * q).sum(1, keepdims=True, dtype=f32)).astype(f32)
        r, a, b, c_ = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        one, two = f32(1.0), f32(2.0)
        R = np.stack([one - two * (b * b + c_ * c_), two * (a * b - r * c_), two * (a * c_ + r * b),
                      two * (a * b + r * c_), one - two * (a * a + c_ * c_), two * (b * c_ - r * a),
                      two * (a * c_ - r * b), two * (b * c_ + r * a), one - two * (a * a + b * b)], 1).reshape(-1, 3, 3).astype(f32)
        M = (R * s[:, None, :]).astype(f32)                     # R S
        S3 = (M @ M.transpose(0, 2, 1)).astype(f32)
    tx, ty, tz = tp(vm, 0), tp(vm, 1), tp(vm, 2)
    limx, limy = f32(1.3) * tfx, f32(1.3) * tfy
    tx = np.minimum(limx, np.maximum(-limx, tx / tz)) * tz
    ty = np.minimum(limy, np.maximum(-limy, ty / tz)) * tz
    fx, fy = f32(W) / (f32(2.0) * tfx), f32(H) / (f32(2.0) * tfy)
    J = np.zeros((len(P), 2, 3), f32)
    J[:, 0, 0], J[:, 0, 2] = fx / tz, -(fx * tx) / (tz * tz)
    J[:, 1, 1], J[:, 1, 2] = fy / tz, -(fy * ty) / (tz * tz)
    Wv = vm[:3, :3].T.astype(f32)                               # world -> view rotation acting on column vectors
    T = (J @ Wv).astype(f32)
    cov = (T @ S3 @ T.transpose(0, 2, 1)).astype(f32)
    ca, cb, cc = cov[:, 0, 0] + f32(0.3), cov[:, 0, 1], cov[:, 1, 1] + f32(0.3)
    det = ca * cc - cb * cb
    inv = f32(1.0) / det
    conic = np.stack([cc * inv, -cb * inv, ca * inv], 1).astype(f32)
    return np.ascontiguousarray(xy), np.ascontiguousarray(conic)


def _sharp_cases():
    import test_gpu_parity as T
    return [c for c in T._fuzz_cases() if c[6] * c[7] < 1.0]


@pytest.mark.parametrize("case", _sharp_cases(), ids=lambda c: f"{c[0]}-N{c[1]}-{c[2]}x{c[3]}-fov{c[5]}-s{c[6]}-m{c[7]}-{c[8]}")
def test_public_binary32_formulation_is_no_more_accurate_on_sub_pixel_splats(case):
    i, N, W, H, deg, fov, sigma, smod, mode, posed = case
    sc = parity.syn.make_scene(N, W, H, sh_degree=deg, seed=100 + i, fovx=fov, sigma_px=sigma, posed=posed)
    sc["scale_modifier"] = smod
    kw = parity.scene_kwargs(sc, mode, bg=(0.1 * (i % 3), 0.5, 1.0 - 0.1 * (i % 5)))
    o = binding.OracleRender(**kw)
    ref = o.forward()[0].astype(np.float64)
    keep = (o.px_ambig == 0)                                    # (pixels on a rounding edge of a discrete decision: either branch is right)
    lib = parity.hostemu_lib()
    ours = parity.hostemu_run(o)["fwd"][0].astype(np.float64)
    geo = o.geom()
    xy, conic = _public_projection_f32(kw)
    rgb = np.ascontiguousarray(geo["rgb"].astype(f32))
    vis = o.forward()[1] > 0
    # sanity: the float32 statement is the same projection (it must agree with the oracle's float64 one to binary32 accuracy)
    assert np.abs(xy[vis] - geo["xy"][vis]).max() <= 2e-3 * max(1.0, np.abs(geo["xy"][vis]).max()) and \
        np.abs(conic[vis] - geo["conic"][vis]).max() <= 2e-3 * max(1.0, np.abs(geo["conic"][vis]).max())
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.hostemu_override_geom.argtypes = [C.c_void_p] * 3
    lib.hostemu_override_geom(p(xy), p(conic), p(rgb))
    lib.hostemu_set_absolute_pixels(1)
    try:
        pub = parity.hostemu_run(o)["fwd"][0].astype(np.float64)
    finally:
        lib.hostemu_set_absolute_pixels(0)
        lib.hostemu_override_geom(None, None, None)
    lib.hostemu_set_no_remainder(1)
    try:
        old = parity.hostemu_run(o)["fwd"][0].astype(np.float64)
    finally:
        lib.hostemu_set_no_remainder(0)
    e_ours = float((np.abs(ours - ref).max(0) * keep).max())
    e_old = float((np.abs(old - ref).max(0) * keep).max())
    e_pub = float((np.abs(pub - ref).max(0) * keep).max())
    print(f"[6b] case {case}: max |image - oracle| on unambiguous pixels: this build's arithmetic {e_ours:.2e} (without the remainder, rounds 1-4: "
          f"{e_old:.2e}), public binary32 formulation {e_pub:.2e}")
    if i == 13:        # the deep stack of non-saturating layers: binary32 accumulation, in every formulation
        assert 1e-5 < e_ours <= 1.4e-5 and e_old <= 1.5e-5 and e_pub > 1e-5
    else:
        assert e_ours <= 1e-6 and e_ours <= e_old + 1e-9
    assert e_pub >= e_ours, (e_ours, e_pub)                     # the public formulation is never the more accurate one
    o.close()
