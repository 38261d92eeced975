"""One-segment-per-GPU sharding of the hierarchical trainer and the merge-step exchange.

The reference trains 2^level leaf segments one after another on cuda:0 and merges neighbours pairwise
(/root/reference/trainer/ht3dgs_trainer.py:710-804; merge_two_3DGS :214-272).  Leaf segments are independent
(:729-753), so here rank r owns leaf segment r; the ONLY data exchange is at a merge, where the source
segment's Gaussian tensors (59 fp32 = 236 B per Gaussian: _xyz, _features_dc, _features_rest, _opacity,
_scaling, _rotation -- :257-267) travel src -> dst point-to-point.  xGMI is a full mesh of point-to-point
links, so each pair of a merge level uses its own link; no ring collective is involved.  Backend "nccl" is
RCCL on ROCm; the same code runs over gloo on CPU for the world_size-2 tests.
"""
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

SEGMENT_KEYS = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
_TRAILING = {"_xyz": (3,), "_features_dc": (1, 3), "_features_rest": (15, 3), "_opacity": (1,), "_scaling": (3,),
             "_rotation": (4,)}


def merge_schedule(world: int) -> List[List[Tuple[int, int]]]:
    """Tree of (dst, src) pairs per merge level: (2k,2k+1), then (4k,4k+2), ... mirroring :767-804 where the
    earlier (even) segment is the destination."""
    assert world >= 1 and (world & (world - 1)) == 0, "number of leaf segments must be a power of two"
    levels, step = [], 1
    while step < world:
        levels.append([(k, k + step) for k in range(0, world, 2 * step)])
        step *= 2
    return levels


def partner(rank: int, level_pairs: List[Tuple[int, int]]) -> Optional[Tuple[str, int]]:
    for dst, src in level_pairs:
        if rank == dst:
            return ("recv", src)
        if rank == src:
            return ("send", dst)
    return None


def send_segment(seg: Dict[str, torch.Tensor], dst: int, extra: Optional[torch.Tensor] = None, group=None) -> None:
    """Count first (N differs per segment), then one flat 59-float-per-Gaussian message."""
    n = seg["_xyz"].shape[0]
    dev = seg["_xyz"].device
    k = 0 if extra is None else extra.numel()
    dist.send(torch.tensor([n, k], dtype=torch.int64, device=dev), dst, group=group)
    flat = torch.cat([seg[key].detach().reshape(n, -1).float() for key in SEGMENT_KEYS], dim=1).contiguous()
    dist.send(flat, dst, group=group)
    if k:
        dist.send(extra.detach().float().contiguous().reshape(-1), dst, group=group)


def recv_segment(src: int, device, group=None) -> Tuple[Dict[str, torch.Tensor], Optional[torch.Tensor]]:
    hdr = torch.zeros(2, dtype=torch.int64, device=device)
    dist.recv(hdr, src, group=group)
    n, k = int(hdr[0].item()), int(hdr[1].item())
    flat = torch.empty((n, 59), dtype=torch.float32, device=device)
    dist.recv(flat, src, group=group)
    seg, o = {}, 0
    for key in SEGMENT_KEYS:
        w = 1
        for d in _TRAILING[key]:
            w *= d
        seg[key] = flat[:, o:o + w].reshape((n,) + _TRAILING[key]).contiguous()
        o += w
    extra = None
    if k:
        extra = torch.empty(k, dtype=torch.float32, device=device)
        dist.recv(extra, src, group=group)
    return seg, extra


def merge_segments(dst_seg: Dict[str, torch.Tensor], src_seg: Dict[str, torch.Tensor], dst_keep: torch.Tensor,
                   src_keep: torch.Tensor, src_to_dst: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """merge_two_3DGS (:233-271): prune both by their importance masks, move the source points by the 4x4
    relative transform, append."""
    out = {}
    xyz = src_seg["_xyz"][src_keep]
    if src_to_dst is not None:
        T = src_to_dst.to(xyz)
        xyz = xyz @ T[:3, :3].t() + T[:3, 3]
    for key in SEGMENT_KEYS:
        s = xyz if key == "_xyz" else src_seg[key][src_keep]
        out[key] = torch.cat([dst_seg[key][dst_keep], s], dim=0)
    return out
