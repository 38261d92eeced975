"""Where the forward blend's time goes, wave by wave (experiment build only).

Needs the library compiled with -DGSR_K6_TIMING (tools/k6_wave_timing.sh does that and restores the default build): every wave
of k_blend_fwd_w6 then records its start / end on the 100 MHz wall clock, HW_ID / XCC_ID and its list length / staged length.
Prints: kernel span, distribution of wave durations, busy time per SIMD (sum of its waves' durations / span), start-time spread.
"""
import ctypes as C
import importlib
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
syn = importlib.import_module("3dgs_hierarchical_training_amd.synthetic")
ts = importlib.import_module("3dgs_hierarchical_training_amd.train_step")
L = importlib.import_module("3dgs_hierarchical_training_amd._lib")

import os
lib = L.load()
for kv in os.environ.get("GSR_OPTS", "").split(","):
    if "=" in kv:
        k_, v_ = kv.split("=")
        print("option", k_, v_, lib.gsr_set_option(k_.encode(), int(v_)))
dev = torch.device("cuda:0")
N, W, H, deg = 1_000_000, 980, 545, 3
scene = syn.make_scene(N, W, H, sh_degree=deg, seed=0)
p = ts.GaussianParams(scene, dev)
st = ts.make_settings(scene, dev, deg)
gt = syn.target_image(W, H, seed=1).to(dev)
for _ in range(4):
    ts.train_step(p, st, gt)      # forward + loss + backward (+ Adam): the last launches of K6 and K8 are what the buffers hold
torch.cuda.synchronize()
raw = C.CDLL(L.LIB_PATH)
T = ((W + 15) // 16) * ((H + 15) // 16)


def report(name, fn, blocks, waves_per_block):
    buf = np.zeros(4 * blocks, dtype=np.uint64)
    rc = fn(buf.ctypes.data_as(C.c_void_p), C.c_int(blocks))
    assert rc == 0, rc
    d = buf.reshape(blocks, 4)
    started = d[d[:, 0] > 0]
    d = d[d[:, 1] > 0]
    t0, t1 = d[:, 0].astype(np.int64), d[:, 1].astype(np.int64)
    hw = (d[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
    xcc = (d[:, 2] >> np.uint64(32)).astype(np.int64) & 0xf
    n = (d[:, 3] >> np.uint64(32)).astype(np.int64)
    staged = (d[:, 3] & np.uint64(0xffffffff)).astype(np.int64)
    base = int(started[:, 0].astype(np.int64).min())
    span = (t1.max() - base) / 100.0   # us
    dur = (t1 - t0) / 100.0
    print(f"== {name}: workgroups that ran to the end {len(d)} of {len(started)} started ({waves_per_block} wave(s) each); span {span:.1f} us (first start -> last end, 100 MHz clock)")
    print("   duration us: mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % (dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max()))
    print("   start us:    p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % tuple(np.percentile((t0 - base) / 100.0, [50, 90, 99, 100])))
    print("   end us:      p10 %.1f  p50 %.1f  p90 %.1f  p99 %.1f" % tuple(np.percentile((t1 - base) / 100.0, [10, 50, 90, 99])))
    print("   list length n: mean %.0f max %d   instances walked: mean %.0f max %d   us per 64 walked: %.2f" % (n.mean(), n.max(), staged.mean(), staged.max(), 64.0 * dur.sum() / max(1, staged.sum())))
    # HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    uniq, inv = np.unique(key, return_inverse=True)
    busy = np.bincount(inv, weights=dur)
    cnt = np.bincount(inv)
    print(f"   SIMDs seen {len(uniq)} (wave 0 of each workgroup)  per SIMD: mean {cnt.mean():.1f} min {cnt.min()} max {cnt.max()}")
    print("   sum of durations per SIMD / span (time-averaged residents): mean %.2f  min %.2f  max %.2f" % ((busy / span).mean(), (busy / span).min(), (busy / span).max()))
    last_end = np.zeros(len(uniq))
    np.maximum.at(last_end, inv, (t1 - base) / 100.0)
    print("   last end per SIMD us: p10 %.1f  p50 %.1f  p90 %.1f  max %.1f" % tuple(np.percentile(last_end, [10, 50, 90, 100])))
    cukey = key // 4
    uc, ic = np.unique(cukey, return_inverse=True)
    cb = np.bincount(ic, weights=dur) * waves_per_block
    print(f"   CUs seen {len(uc)}  resident waves per CU (time-averaged): mean {(cb / span).mean():.1f} min {(cb / span).min():.1f} max {(cb / span).max():.1f}")
    xb = np.bincount(xcc, weights=dur)
    print("   per XCD: workgroups", np.bincount(xcc).tolist(), " busy-sum/span", [round(float(x / span), 1) for x in xb])
    A = np.vstack([staged, np.ones_like(staged)]).T.astype(np.float64)
    coef, *_ = np.linalg.lstsq(A, dur, rcond=None)
    print("   fit duration = %.4f us * walked + %.2f us ; corr %.3f" % (coef[0], coef[1], np.corrcoef(staged, dur)[0, 1]))
    # occupancy over time: residents at 10 us marks
    marks = np.arange(0, span, 10.0)
    occ = [int((((t0 - base) / 100.0 <= m) & ((t1 - base) / 100.0 > m)).sum()) for m in marks]
    print("   resident workgroups at 0, 10, 20, ... us:", occ)


nb6 = min(65536, 8 * 4 * ((T + 7) // 8 + 8))
cnt = np.zeros(2 * nb6, dtype=np.uint32)
assert raw.gsr_debug_k6_counts(cnt.ctypes.data_as(C.c_void_p), C.c_int(nb6)) == 0
cnt = cnt.reshape(nb6, 2).astype(np.int64)
print("K6 visits (reach bit set) %d, of them taken by at least one pixel %d (%.1f %%)" % (cnt[:, 0].sum(), cnt[:, 1].sum(), 100.0 * cnt[:, 1].sum() / max(1, cnt[:, 0].sum())))
cyc = np.zeros(2 * nb6, dtype=np.uint64)
assert raw.gsr_debug_k6_cycles(cyc.ctypes.data_as(C.c_void_p), C.c_int(nb6)) == 0
cyc = cyc.reshape(nb6, 2).astype(np.float64)
live = cyc.sum(1) > 0
print("K6 per wave, s_memtime ticks: staging (wait for the gather, box test, LDS writes) mean %.0f, visit loops mean %.0f -> staging share %.1f %%" % (
    cyc[live, 0].mean(), cyc[live, 1].mean(), 100.0 * cyc[live, 0].sum() / cyc[live].sum()))
report("K6 forward blend (k_blend_fwd_w6)", raw.gsr_debug_k6_timing, min(65536, 8 * 4 * ((T + 7) // 8 + 8)), 1)
report("K8 backward blend (k_blend_bwd2)", raw.gsr_debug_k8_timing, 65536, 2)
