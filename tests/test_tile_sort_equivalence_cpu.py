"""The argument behind the tile-sort route (csrc/gsr_kernels.hip k_tile_sort*), as a property of plain arrays: the
reference's pipeline orders ALL (tile, Gaussian) instances by (tile, depth) with a stable sort of Gaussians that arrive in index order
(the duplicateWithKeys + radix sort of the public rasterizer; here: depth sort of the Gaussians, then per-tile lists in that order).
Building every tile's list in INDEX order first and sorting each list by depth key -- stably -- must give the same lists, ties included.
No GPU, no library: numpy only."""
import numpy as np
import pytest


def _lists_global_depth_order(keys, touches, T):
    order = np.argsort(keys, kind="stable")               # stable sort of (key, index): equal keys stay in index order
    return [[int(g) for g in order if t in touches[g]] for t in range(T)]


def _lists_index_order_then_tile_sort(keys, touches, T):
    out = []
    for t in range(T):
        seg = np.array([g for g in range(len(keys)) if t in touches[g]], dtype=np.int64)   # what the index-order binning places
        if seg.size:
            seg = seg[np.argsort(keys[seg], kind="stable")]                                # what k_tile_sort does to the segment
        out.append([int(g) for g in seg])
    return out


@pytest.mark.parametrize("seed,n,T,levels", [(0, 200, 7, 5), (1, 500, 12, 3), (2, 64, 1, 2), (3, 300, 9, 1000), (4, 1, 3, 1)])
def test_per_tile_stable_sort_of_index_ordered_lists_is_the_global_order(seed, n, T, levels):
    rng = np.random.default_rng(seed)
    keys = rng.integers(0, levels, size=n).astype(np.uint32)          # few levels: many equal depth keys
    touches = [set(rng.choice(T, size=rng.integers(0, min(T, 4) + 1), replace=False).tolist()) for _ in range(n)]
    a = _lists_global_depth_order(keys, touches, T)
    b = _lists_index_order_then_tile_sort(keys, touches, T)
    assert a == b


def test_lsd_digit_passes_with_skipped_uniform_digits_sort_stably():
    """k_tile_sort's passes: 8-bit digits from the low end, a digit that every key of the segment shares is skipped.  Same result as one
    stable sort of the whole keys."""
    rng = np.random.default_rng(7)
    keys = np.uint32(0x3F000000) | (rng.integers(0, 256, size=777).astype(np.uint32) << np.uint32(16)) | rng.integers(0, 256, size=777).astype(np.uint32)   # bytes 1 and 3 uniform
    vals = np.arange(keys.size)
    differ = np.bitwise_or.reduce(keys) & ~np.bitwise_and.reduce(keys)
    k, v = keys.copy(), vals.copy()
    skipped = 0
    for shift in (0, 8, 16, 24):
        if (int(differ) >> shift) & 0xFF == 0:
            skipped += 1
            continue
        o = np.argsort((k >> np.uint32(shift)) & np.uint32(0xFF), kind="stable")
        k, v = k[o], v[o]
    assert skipped == 2
    ref = np.argsort(keys, kind="stable")
    assert np.array_equal(v, vals[ref])
