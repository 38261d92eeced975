"""GPU: tile lists cut where the tile stopped at the frame's previous render (round 6, include/gsr.h "list_cut").

The cut changes which list words are WRITTEN, never what a pixel blends: counts, scans, tile bases, R and every pair's position
are the full binning's, a tile's list is written up to a chunk boundary of the depth order behind the depth its waves needed last
time, and a wave that runs out of a cut list with a live pixel has its tile repaired ON THE DEVICE (scatter of the left-out chunks
for the flagged tiles, blend of the flagged tiles on their full lists).  So everything below is an EQUALITY: image, depth, alpha,
radii, R, the tiles' range starts, the list's valid prefixes, and -- with the backward's fixed-order accumulation
("deterministic_backward") -- every gradient, bit for bit, between `list_cut` 1 and 0; on a scene that does not change (no repair),
on one that changes under the same frame so that tiles need MORE than they kept (repairs), with large rects, with a speculative
capacity that overflows, with another model under the same frame.  Plus: the oracle parity of a cut render."""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch

import parity
from oracle import binding

pytestmark = pytest.mark.gpu
L = importlib.import_module("3dgs_hierarchical_training_amd._lib")
R_ = importlib.import_module("3dgs_hierarchical_training_amd.rasterizer")


def _stats(lib, W, H):
    out = (C.c_int64 * 5)()
    assert lib.gsr_debug_list_cut_stats(W, H, out) == 0
    return [int(v) for v in out]


def _render(kw, grads=None):
    import hip_runner
    out = hip_runner.run_hip(kw, grads)
    ranges, lst = R_.last_binning()
    info = R_.last_call_info()
    return out, ranges.cpu().numpy(), lst.cpu().numpy(), info["num_rendered"], info["staged"]


def _equal_renders(a, b, what):
    (oa, ra, la, Ra, sa), (ob, rb, lb, Rb, sb) = a, b
    assert Ra == Rb and sa == sb, (what, Ra, Rb, sa, sb)
    for x, y, name in zip(oa["fwd"], ob["fwd"], ("color", "radii", "depth", "alpha")):
        assert np.array_equal(x, y), (what, name, float(np.abs(x.astype(np.float64) - y).max()))
    # the cut side's ranges are prefixes of the full side's, list contents equal on them
    full_r, cut_r, full_l, cut_l = (rb, ra, lb, la) if (ra[:, 1] - ra[:, 0]).sum() <= (rb[:, 1] - rb[:, 0]).sum() else (ra, rb, la, lb)
    nz = cut_r[:, 1] > cut_r[:, 0]
    assert np.array_equal(cut_r[nz, 0], full_r[nz, 0]) and np.all(cut_r[nz, 1] <= full_r[nz, 1]), what
    for t in np.nonzero(nz)[0][:: max(1, int(nz.sum()) // 400)]:      # (a sample of tiles: slicing 2 000 lists in numpy is slow)
        x, y = int(cut_r[t, 0]), int(cut_r[t, 1])
        assert np.array_equal(cut_l[x:y], full_l[x:y]), (what, int(t))
    if "grads" in oa:
        for k in oa["grads"]:
            assert np.array_equal(oa["grads"][k], ob["grads"][k]), (what, k, float(np.abs(oa["grads"][k] - ob["grads"][k]).max()))


def _two_ways(lib, kws, W, H, grads, margin=None, hint=None):
    """Render the sequence of scenes `kws` (the same frame: same camera, same image size) with the cut on and off, from a clean cache;
    returns ([cut renders], [full renders], stats of the cut run)."""
    res = {}
    st = None
    assert lib.gsr_set_option(b"deterministic_backward", 1) == 0
    # (models of these sizes take the tile-sort route by default, which has no global depth order to cut along: the cut serves the
    #  depth-sort route that large models take -- forced here, so that the seconds-sized scenes of a test exercise it)
    assert lib.gsr_set_option(b"tile_sort", 0) == 0
    try:
        for cut in (1, 0):
            assert lib.gsr_set_option(b"list_cut", 2 if cut else 0) == 0       # (2: whatever the model's size)
            assert lib.gsr_set_option(b"list_cut_margin_e3", -1 if margin is None else margin) == 0
            assert lib.gsr_set_option(b"view_cache_reset", 1) == 0
            assert lib.gsr_set_option(b"reset_speculation", 1) == 0
            outs = []
            for i, kw in enumerate(kws):
                if hint is not None and i == hint[0]:
                    assert lib.gsr_set_option(b"binning_capacity_hint", hint[1]) == 0
                outs.append(_render(kw, grads))
            res[cut] = outs
            if cut:
                st = _stats(lib, W, H)
    finally:
        lib.gsr_set_option(b"list_cut", 1)
        lib.gsr_set_option(b"tile_sort", 1)
        lib.gsr_set_option(b"list_cut_margin_e3", -1)
        lib.gsr_set_option(b"deterministic_backward", 0)
        lib.gsr_set_option(b"binning_capacity_hint", 0)
    return res[1], res[0], st


CASES = [(20_000, 320, 240, 3, 1.0), (130_000, 980, 545, 0, 1.0), (300_000, 980, 545, 3, 1.0), (60_000, 640, 360, 1, 6.0)]


@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}@{c[1]}x{c[2]}-deg{c[3]}-x{c[4]}" for c in CASES])
def test_cut_lists_render_what_full_lists_render(case):
    """A frame rendered four times (the first sight of it is never cut): with the cut on, the later renders write shorter lists --
    chunks of the depth order are skipped -- and everything a caller can observe is what full lists give, bit for bit; no tile
    needs a repair on a scene that does not change."""
    N, W, H, deg, smod = case
    lib = L.load()
    sc = parity.syn.make_scene(N, W, H, sh_degree=deg, seed=N % 89, posed=True)
    sc["scale_modifier"] = smod                     # (x6: rects of more than 32 tiles, walked by whole waves)
    kw = parity.scene_kwargs(sc, "sh", bg=(0.1, 0.2, 0.3))
    gc, gd, ga = parity.upstream_grads(H, W, seed=5)
    cut, full, st = _two_ways(lib, [kw] * 4, W, H, (gc, gd, ga))
    for i, (a, b) in enumerate(zip(cut, full)):
        _equal_renders(a, b, f"render {i}")
    assert st[0] == 4 and st[1] == 0 and st[2] == 0, st
    kept = [(r[1][:, 1] - r[1][:, 0]).sum() for r in cut]
    assert kept[0] == cut[0][3]                       # the first render of a frame: full lists (R)
    assert kept[1] <= kept[0] and kept[2] == kept[1] and kept[3] == kept[1] and kept[1] >= cut[0][4], (st, kept, cut[0][3], cut[0][4])
    if cut[0][4] < 0.5 * cut[0][3]:                   # tiles that saturate early: whole chunks of the depth order are left out
        assert st[3] > 0 and kept[1] < 0.8 * kept[0], (st, kept, cut[0][3], cut[0][4])
    print(f"[list cut] {case}: R {cut[0][3]}, written {kept[1]} ({kept[1] / max(1, cut[0][3]):.1%}), staged {cut[0][4]}; chunks skipped (3 renders) {st[3]}")


def test_tiles_that_need_more_than_they_kept_are_repaired_on_the_device():
    """The same frame, the same model size, but between two renders the model turns transparent (opacities x 0.15): tiles now need
    far more of their lists than they kept -- with NO margin.  The waves that run out of
    their cut lists flag their tiles and the repair pass re-blends those on full lists: everything equals the full-list render,
    the counters say that repairs happened, and the render after that keeps what the repaired tiles needed (no repair again)."""
    lib = L.load()
    N, W, H = 80_000, 640, 360
    sc = parity.syn.make_scene(N, W, H, sh_degree=1, seed=31, posed=True)
    sc["scale_modifier"] = 4.0                      # (dense: tiles saturate long before their lists end)
    kw0 = parity.scene_kwargs(sc, "sh", bg=(0.0, 0.1, 0.0))
    sc1 = dict(sc)
    sc1["opacities"] = sc["opacities"] * 0.15
    kw1 = parity.scene_kwargs(sc1, "sh", bg=(0.0, 0.1, 0.0))
    gc, gd, ga = parity.upstream_grads(H, W, seed=6)
    cut, full, st = _two_ways(lib, [kw0, kw0, kw1, kw1, kw1], W, H, (gc, gd, ga), margin=0)
    for i, (a, b) in enumerate(zip(cut, full)):
        _equal_renders(a, b, f"render {i}")
    assert st[0] == 5 and st[1] >= 1 and st[2] > 50, st          # render 2 ran into its cuts all over the frame
    assert cut[2][4] > 1.5 * cut[1][4]                             # (the transparent model stages far deeper)
    cut2, full2, st2 = _two_ways(lib, [kw0, kw0, kw1, kw1], W, H, None, margin=0)
    cut3, full3, st3 = _two_ways(lib, [kw0, kw0, kw1], W, H, None, margin=0)
    assert st2[1] == st3[1] and st2[2] == st3[2], (st2, st3)      # the render AFTER the repaired one needed no repair
    print(f"[list cut] repairs: {st[1]} renders, {st[2]} tiles; staged {cut[1][4]} -> {cut[2][4]}")


def test_cut_with_an_overflowing_speculative_capacity():
    """A cut render whose speculative binning capacity is too small is run again on the exact size -- the cut tables are the same,
    the flags of the truncated first attempt must not leak into the result."""
    lib = L.load()
    N, W, H = 50_000, 480, 320
    sc = parity.syn.make_scene(N, W, H, sh_degree=0, seed=7, posed=True)
    kw = parity.scene_kwargs(sc, "sh")
    gc, gd, ga = parity.upstream_grads(H, W, seed=1)
    cut, full, st = _two_ways(lib, [kw] * 4, W, H, (gc, gd, ga), hint=(2, 20_000))
    for i, (a, b) in enumerate(zip(cut, full)):
        _equal_renders(a, b, f"render {i}")
    assert lib.gsr_get_counter(b"spec_overflows") >= 1


def test_another_model_under_the_same_frame_is_not_cut_by_the_first_ones_depths():
    """Teacher and student, a stage-A model next to a leaf: two models of different size rendered alternately through the same
    camera.  The remembered depths belong to ONE model size; the other model's renders must not be cut by them (they would be
    repaired, but every render of the pair would pay for it)."""
    lib = L.load()
    W, H = 480, 320
    a = parity.scene_kwargs(parity.syn.make_scene(40_000, W, H, sh_degree=0, seed=3, posed=True), "sh")
    b = parity.scene_kwargs(parity.syn.make_scene(9_000, W, H, sh_degree=0, seed=4, posed=True), "sh")
    for k in ("viewmatrix", "projmatrix", "campos"):
        b[k] = a[k]
    cut, full, st = _two_ways(lib, [a, b, a, b, a, b], W, H, None)
    for i, (x, y) in enumerate(zip(cut, full)):
        _equal_renders(x, y, f"render {i}")
    assert st[1] == 0 and st[2] == 0 and st[3] == 0, st          # alternating sizes: never cut (and never a repair)


def test_oracle_parity_of_a_cut_render():
    """The parity test proper on a render that ran with cut lists (the third of its frame): forward against the float64 oracle with
    the rounding-edge resolution, gradients at the usual bars."""
    import hip_runner
    lib = L.load()
    N, W, H = 30_000, 400, 300
    sc = parity.syn.make_scene(N, W, H, sh_degree=3, seed=77, posed=True)
    sc["scale_modifier"] = 4.0                      # (dense: the cut leaves whole chunks of the depth order out)
    kw = parity.scene_kwargs(sc, "sh", bg=(0.3, 0.2, 0.1))
    assert lib.gsr_set_option(b"view_cache_reset", 1) == 0
    assert lib.gsr_set_option(b"tile_sort", 0) == 0
    assert lib.gsr_set_option(b"list_cut", 2) == 0
    try:
        hip_runner.run_hip(kw); hip_runner.run_hip(kw)
        s0 = _stats(lib, W, H)
        o = binding.OracleRender(**kw)
        rep, out, ref = parity.oracle_case(o, lambda g: hip_runner.run_hip(kw, g), parity.upstream_grads(H, W, seed=2), "cut render")
        parity.check_grads(out["grads"], ref, "cut render")
        s1 = _stats(lib, W, H)
    finally:
        lib.gsr_set_option(b"tile_sort", 1)
        lib.gsr_set_option(b"list_cut", 1)
    assert s1[0] - s0[0] == 2 and s1[3] > s0[3] and s1[1] == s0[1]
    o.close()
