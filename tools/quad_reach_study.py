#!/usr/bin/env python3
"""How much of a blend visit is wasted on pixels the instance cannot reach -- and what a finer-grained walk would save.

For a rendered frame this script takes the library's own tile lists (gsr_debug_read_binning), splat records and per-sub-tile staged
depths, evaluates the kernels' conservative box test (gsr_math.h box_accept) on every (instance, 4x4 pixel quad) and (instance, 8x8
sub-tile), and counts, over the instances each 8x8 wave actually stages:
  wave visits      = what k_blend_fwd_w6 does today: one visit per instance whose box test passes on the 8x8 block
  row visits       = visits if each 16-lane DPP row (one 4x4 quad) walked ITS OWN instances and the wave lasted as long as its
                     slowest row, rows re-synchronised at every batch of 64 staged instances / never
  ideal            = sum over quads / 4 (perfect balance)
and the same for the backward's geometry (two pixels per lane: a wave = 16x8 pixels, a row = 8x4 or 4x8 pixels).
Run on the GPU box: python tools/quad_reach_study.py [headline|stage_a|stage_a_pixel|c3]"""
import ctypes as C
import importlib
import math
import sys

import torch

sys.path.insert(0, ".")
syn = importlib.import_module("3dgs_hierarchical_training_amd.synthetic")
ts = importlib.import_module("3dgs_hierarchical_training_amd.train_step")
L = importlib.import_module("3dgs_hierarchical_training_amd._lib")
E = importlib.import_module("3dgs_hierarchical_training_amd._ext")
lib = L.load()
ops = E.load()
dev = torch.device("cuda:0")


def box_accept(px, py, ca, cb, cc, op, bx0, by0, bx1, by1):
    tau = torch.where(op > 0, 2.0 * torch.log(255.0 * op.clamp_min(1e-30)), torch.full_like(op, -1.0))
    slack = 1e-3 * (1.0 + tau.abs())
    hi = tau + slack
    none = tau < -slack
    dx0, dx1, dy0, dy1 = bx0 - px, bx1 - px, by0 - py, by1 - py
    inside = (dx0 <= 0) & (dx1 >= 0) & (dy0 <= 0) & (dy1 >= 0)
    rc, ra = cb / cc, cb / ca
    q = lambda x, y: ca * x * x + 2 * cb * x * y + cc * y * y
    y = torch.minimum(dy1, torch.maximum(dy0, -rc * dx0)); qmin = q(dx0, y)
    y = torch.minimum(dy1, torch.maximum(dy0, -rc * dx1)); qmin = torch.minimum(qmin, q(dx1, y))
    x = torch.minimum(dx1, torch.maximum(dx0, -ra * dy0)); qmin = torch.minimum(qmin, q(x, dy0))
    x = torch.minimum(dx1, torch.maximum(dx0, -ra * dy1)); qmin = torch.minimum(qmin, q(x, dy1))
    return (~none) & (inside | (qmin <= hi))


def study(name, scene, deg, p=None):
    W, H = int(scene["image_width"]), int(scene["image_height"])
    p = p or ts.GaussianParams(scene, dev)
    st = ts.make_settings(scene, dev, deg)
    e = torch.empty(0, device=dev)
    eb = torch.empty(0, dtype=torch.uint8, device=dev)
    with torch.no_grad():
        out = ops.rasterize_forward(p._xyz, p._features_dc, e, p._opacity, p._scaling, p._rotation, e, p._features_rest, st.viewmatrix, st.projmatrix,
                                    st.campos, st.bg, e, H, W, float(st.tanfovx), float(st.tanfovy), 1.0, deg, True, False, False, eb, [], 0)
    color, radii, depth, alpha, geom, image, binning, meta = out
    R, cap = int(meta[0]), int(meta[1])
    tiles_x, tiles_y = (W + 15) // 16, (H + 15) // 16
    T = tiles_x * tiles_y
    ranges = torch.zeros(T, 2, dtype=torch.int32, device=dev)
    lst = torch.zeros(max(R, 1), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    assert lib.gsr_debug_read_binning(C.c_void_p(binning.data_ptr()), cap, R, W, H, C.c_void_p(ranges.data_ptr()), C.c_void_p(lst.data_ptr()), C.c_void_p(stream)) == 0
    torch.cuda.synchronize()
    N = p._xyz.shape[0]
    sp = geom[:48 * N].view(torch.float32).view(N, 12)
    so = lib.gsr_image_staged_offset(W, H)
    staged = image[so:so + 16 * T].view(torch.int32).view(T, 4).long()
    ranges = ranges.long()
    n_t = ranges[:, 1] - ranges[:, 0]
    tile_of = torch.repeat_interleave(torch.arange(T, device=dev), n_t)          # tile of every list entry (lists are contiguous in tile order)
    assert tile_of.numel() == R, (tile_of.numel(), R)
    # position inside the tile's list
    start = torch.repeat_interleave(ranges[:, 0], n_t)
    order = torch.argsort(start, stable=True)      # lists are laid out by tile base; entries of a tile are consecutive from ranges[t].x
    # (the list is [ranges[t].x, ranges[t].y) per tile: entry index = ranges[t].x + k)
    k = torch.arange(R, device=dev) - torch.repeat_interleave(torch.cumsum(n_t, 0) - n_t, n_t)
    idx = start + k
    gid = lst.long()[idx]
    s = sp[gid]
    px, py, ca, cb, cc, op = s[:, 0], s[:, 1], s[:, 2], s[:, 3], s[:, 4], s[:, 5]
    tx, ty = (tile_of % tiles_x).float(), (tile_of // tiles_x).float()
    cx, cy = 0.5 * W, 0.5 * H
    res = {}

    def reach(x0, y0, w, h):
        bx0 = tx * 16 + x0 - cx
        by0 = ty * 16 + y0 - cy
        bx1 = torch.minimum(bx0 + (w - 1), torch.full_like(bx0, (W - 1) - cx))
        by1 = torch.minimum(by0 + (h - 1), torch.full_like(by0, (H - 1) - cy))
        ok = box_accept(px, py, ca, cb, cc, op, bx0, by0, bx1, by1)
        return ok & (bx0 <= (W - 1) - cx) & (by0 <= (H - 1) - cy)

    batch = k // 64
    nbat = int(batch.max().item()) + 1 if R else 1

    def per_wave(sub_boxes, row_boxes_of_sub, label):
        """sub_boxes: [(x0, y0, w, h)] of the waves of a tile; row_boxes_of_sub[i]: the four row boxes of wave i."""
        wave_vis = 0
        row_sync, row_free, ideal, lanes_live = 0, 0, 0.0, 0.0
        for i, (x0, y0, w, h) in enumerate(sub_boxes):
            # which staged counter applies: the forward's sub-tile(s) covered by this wave -- an instance is walked while ANY of them still stages it
            subs = [(yy // 8) * 2 + (xx // 8) for yy in range(y0, y0 + h, 8) for xx in range(x0, x0 + w, 8)]
            lim = torch.stack([staged[:, sidx] for sidx in subs], 1).max(1).values
            live = k < lim[tile_of]
            r_w = reach(x0, y0, w, h) & live
            wave_vis += int(r_w.sum())
            rows = torch.stack([reach(*rb) & live for rb in row_boxes_of_sub[i]], 1)          # [R, 4]
            key = tile_of * nbat + batch
            cnt = torch.zeros(T * nbat, 4, device=dev)
            cnt.index_add_(0, key, rows.float())
            row_sync += int(cnt.max(1).values.sum())
            cnt_t = torch.zeros(T, 4, device=dev)
            cnt_t.index_add_(0, tile_of, rows.float())
            row_free += int(cnt_t.max(1).values.sum())
            ideal += float(cnt_t.sum()) / 4.0
        res[label] = {"wave_visits": wave_vis, "row_visits_sync_per_64": row_sync, "row_visits_free": row_free, "row_visits_ideal": int(ideal),
                      "gain_sync": wave_vis / max(1, row_sync), "gain_free": wave_vis / max(1, row_free), "gain_ideal": wave_vis / max(1.0, ideal)}
    # forward: wave = 8x8, row = 4x4 quad (lane = y * 8 + x would put a row on 2 lines of 8: the kernel would use a quad-major lane map)
    fsub = [(0, 0, 8, 8), (8, 0, 8, 8), (0, 8, 8, 8), (8, 8, 8, 8)]
    per_wave(fsub, [[(x0 + qx, y0 + qy, 4, 4) for qy in (0, 4) for qx in (0, 4)] for (x0, y0, _, _) in fsub], "fwd 8x8 wave, rows = 4x4 quads")
    per_wave(fsub, [[(x0, y0 + 2 * r, 8, 2) for r in range(4)] for (x0, y0, _, _) in fsub], "fwd 8x8 wave, rows = 8x2 strips (today's lane map)")
    # backward: wave = 16x8 (two pixels per lane), rows = 8x4 blocks / 16x2 strips
    bsub = [(0, 0, 16, 8), (0, 8, 16, 8)]
    per_wave(bsub, [[(qx, y0 + qy, 8, 4) for qy in (0, 4) for qx in (0, 8)] for (_, y0, _, _) in bsub], "bwd 16x8 wave, rows = 8x4 blocks")
    per_wave(bsub, [[(qx, y0 + qy, 4, 8) for qy in (0,) for qx in (0, 4, 8, 12)] for (_, y0, _, _) in bsub], "bwd 16x8 wave, rows = 4x8 blocks")
    print(f"== {name}: N {N}, {W}x{H}, R {R}, staged {int(staged.sum() // 4)} per sub-tile column")
    for kx, v in res.items():
        print(f"   {kx}: wave visits {v['wave_visits']:,}  row visits sync/64 {v['row_visits_sync_per_64']:,} (x{v['gain_sync']:.2f})  "
              f"free {v['row_visits_free']:,} (x{v['gain_free']:.2f})  ideal {v['row_visits_ideal']:,} (x{v['gain_ideal']:.2f})")
    return res


which = sys.argv[1:] or ["headline", "stage_a", "stage_a_pixel", "c3"]
if "headline" in which:
    study("headline 1M @980x545", syn.make_scene(1_000_000, 980, 545, sh_degree=3, seed=0), 3)
if "stage_a" in which:
    study("stage-A size 130k random @980x545", syn.make_scene(130_000, 980, 545, sh_degree=0, seed=3), 0)
if "stage_a_pixel" in which:
    sequence = importlib.import_module("3dgs_hierarchical_training_amd.sequence")
    seq = sequence.FrameSequence(2, 400_000, 980, 545, dev, seed=0)
    sc = seq.pixel_scene(0, stride=2, seed=0)
    study("stage-A pixel-Gaussians (stride 2) @980x545", sc, 0)
if "c3" in which:
    study("C3 1M @1920x1080", syn.make_scene(1_000_000, 1920, 1080, sh_degree=3, seed=0), 3)
