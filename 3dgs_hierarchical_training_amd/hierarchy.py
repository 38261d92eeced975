"""Merge step of the hierarchical trainer on top of the MI355X rasterizer.

Mirrors, without importing the reference:
  * `HTGaussianTrainer.calc_importance`  /root/reference/trainer/ht3dgs_trainer.py:1427-1462 -- per view,
    loss = rendering.sum() on the clamped image, |grad| of `_features_dc` / `_features_rest` accumulated over the
    views (the `register_hook(lambda g: g.abs())` of :1436-1437), divided by the number of pixels;
  * `merge_two_3DGS`                     :214-272 -- amax over the 48 coefficients, `topk(largest=False)` of
    `prune_ratio * N` marks the Gaussians to drop on BOTH sides, source points moved by the 4x4, append.
Multi-GPU: the source rank prunes with its own importance and ships only the survivors (59 floats each)
point-to-point to the destination rank (segments.py); no collective is involved.
"""
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import segments
from .rasterizer import GaussianRasterizationSettings, rasterize_gaussians_raw


def calc_importance(seg: Dict[str, torch.Tensor], views: List[GaussianRasterizationSettings]) -> torch.Tensor:
    """[N, 48] colour importance of a segment over `views` (each a full raster-settings tuple)."""
    dc = seg["_features_dc"].detach().clone().requires_grad_(True)
    rest = seg["_features_rest"].detach().clone().requires_grad_(True)
    acc_dc, acc_rest = torch.zeros_like(dc), torch.zeros_like(rest)
    num_pixels = 0
    for rs in views:
        dc.grad = rest.grad = None
        m2d = torch.zeros_like(seg["_xyz"])
        color = rasterize_gaussians_raw(seg["_xyz"].detach(), m2d, dc, rest, seg["_opacity"].detach(),
                                        seg["_scaling"].detach(), seg["_rotation"].detach(), rs)[0]
        color.clamp(0, 1).sum().backward()
        acc_dc += dc.grad.abs()
        acc_rest += rest.grad.abs()
        num_pixels += int(rs.image_height) * int(rs.image_width)
    return (torch.cat([acc_dc, acc_rest], 1).flatten(-2) / max(num_pixels, 1)).detach()


def prune_mask(importance: torch.Tensor, prune_ratio: float) -> torch.Tensor:
    """True = dropped: the `prune_ratio * N` Gaussians with the smallest max-over-coefficients importance
    (ht3dgs_trainer.py:234-239)."""
    score = importance.amax(-1).reshape(-1)
    k = int(score.shape[0] * prune_ratio)
    mask = torch.zeros_like(score, dtype=torch.bool)
    if k > 0:
        mask[torch.topk(score, k, largest=False).indices] = True
    return mask


def merge_level(seg: Dict[str, torch.Tensor], views: List[GaussianRasterizationSettings], level_pairs, prune_ratio: float,
                src_to_dst: Optional[torch.Tensor] = None, importance_fn=calc_importance, group=None):
    """One level of the merge tree for this rank.  Returns the merged segment on destination ranks, None on
    source ranks (their GPU is free after the send), and the unchanged segment on ranks idle at this level."""
    rank = dist.get_rank(group)
    role = segments.partner(rank, level_pairs)
    if role is None:
        return seg
    drop = prune_mask(importance_fn(seg, views), prune_ratio)
    keep = ~drop
    if role[0] == "send":
        segments.send_segment({k: seg[k][keep] for k in segments.SEGMENT_KEYS}, role[1], group=group)
        return None
    got, _ = segments.recv_segment(role[1], seg["_xyz"].device, group=group)
    all_src = torch.ones(got["_xyz"].shape[0], dtype=torch.bool, device=got["_xyz"].device)
    return segments.merge_segments(seg, got, keep, all_src, src_to_dst)
