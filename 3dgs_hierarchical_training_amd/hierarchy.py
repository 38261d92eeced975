"""Merge step of the hierarchical trainer on top of the MI355X rasterizer.

Mirrors, without importing the reference:
  * `HTGaussianTrainer.calc_importance`  /root/reference/trainer/ht3dgs_trainer.py:1427-1462 -- per view,
    loss = rendering.sum() on the clamped image, |grad| of `_features_dc` / `_features_rest` accumulated over the
    views (the `register_hook(lambda g: g.abs())` of :1436-1437), divided by the number of pixels;
  * `merge_two_3DGS`                     :214-272 -- amax over the 48 coefficients, `topk(largest=False)` of
    `prune_ratio * N` marks the Gaussians to drop on BOTH sides, source points moved by the 4x4, append.
Multi-GPU (segments.py): each child's importance is computed on its home rank (both ranks of a pair work in
parallel), the source ships its **un-pruned** child with the drop mask, and the mask is applied at the destination --
which therefore also holds both un-pruned children as the frozen teachers of `train_nonleaf_3DGS_phase1`
(:757, :866-883).  No collective is involved.
"""
import time
from typing import Dict, List, Optional

import torch

from . import segments
from .rasterizer import GaussianRasterizationSettings, rasterize_gaussians_raw


def _elapsed_ms(t0: float, ref: torch.Tensor) -> float:
    if ref.is_cuda:
        torch.cuda.synchronize(ref.device)
    return 1e3 * (time.perf_counter() - t0)


def render_raw(seg: Dict[str, torch.Tensor], rs: GaussianRasterizationSettings) -> torch.Tensor:
    """Clamped image of a frozen model given by its six raw tensors (the teacher render of :877-883)."""
    with torch.no_grad():
        m2d = torch.zeros_like(seg["_xyz"])
        color = rasterize_gaussians_raw(seg["_xyz"], m2d, seg["_features_dc"], seg["_features_rest"], seg["_opacity"],
                                        seg["_scaling"], seg["_rotation"], rs)[0]
        return color.clamp(0, 1)


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


def merge_send(tr, dst: int, seg: Dict[str, torch.Tensor], views, prune_ratio: float, frames=None, poses=None,
               start_fidx: int = 0, global_iteration: int = 0, importance_fn=calc_importance, drop=None, sh_degree: int = -1) -> Dict:
    """Source side of one pair: own importance -> drop mask; ship the UN-PRUNED child + mask + frames / poses.
    `drop`: a mask computed beforehand (then `views` is not used)."""
    t0 = time.perf_counter()
    if drop is None:
        drop = prune_mask(importance_fn(seg, views), prune_ratio)
    imp_ms = _elapsed_ms(t0, seg["_xyz"])
    st = segments.send_child(tr, dst, seg, drop=drop, frames=frames, poses=poses, start_fidx=start_fidx,
                             global_iteration=global_iteration, sh_degree=sh_degree)
    return {"role": "src", "peer": dst, "importance_ms": imp_ms, "send_ms": st["ms"], "bytes": st["bytes"],
            "n": int(seg["_xyz"].shape[0]), "n_dropped": int(drop.sum())}


def merge_recv(tr, src: int, seg: Dict[str, torch.Tensor], views, prune_ratio: float, src_to_dst=None,
               importance_fn=calc_importance, drop=None) -> Dict:
    """Destination side: own importance (in parallel with the source's), receive the un-pruned child, apply both masks,
    move the child's points by `src_to_dst` (a [4,4] or a callable(child_message) -> [4,4]) and append.

    Returns {'merged', 'teachers': [own un-pruned, child un-pruned], 'child': message, stats...}."""
    t0 = time.perf_counter()
    drop_dst = drop if drop is not None else prune_mask(importance_fn(seg, views), prune_ratio)
    imp_ms = _elapsed_ms(t0, seg["_xyz"])
    msg = segments.recv_child(tr, src, seg["_xyz"].device)
    child = msg["seg"]
    drop_src = msg["drop"] if msg["drop"] is not None else torch.zeros(child["_xyz"].shape[0], dtype=torch.bool,
                                                                         device=child["_xyz"].device)
    T = src_to_dst(msg) if callable(src_to_dst) else src_to_dst
    t1 = time.perf_counter()
    merged = segments.merge_segments(seg, child, ~drop_dst, ~drop_src, T)
    own = {k: seg[k].detach() for k in segments.SEGMENT_KEYS}
    return {"role": "dst", "peer": src, "merged": merged, "teachers": [own, child], "child": msg,
            "importance_ms": imp_ms, "recv_ms": msg["ms"], "bytes": msg["bytes"],
            "append_ms": _elapsed_ms(t1, seg["_xyz"]), "n": int(seg["_xyz"].shape[0]), "n_child": int(child["_xyz"].shape[0]),
            "n_merged": int(merged["_xyz"].shape[0]), "n_dropped": int(drop_dst.sum()), "n_child_dropped": int(drop_src.sum())}


def merge_level(seg: Dict[str, torch.Tensor], views: List[GaussianRasterizationSettings], level_pairs, prune_ratio: float,
                src_to_dst: Optional[torch.Tensor] = None, importance_fn=calc_importance, group=None, transport=None):
    """One level of the merge tree for this rank.  Returns the merged segment on destination ranks, None on
    source ranks (their GPU is free after the send), and the unchanged segment on ranks idle at this level."""
    tr = transport if transport is not None else segments.DistTransport(group)
    role = segments.partner(tr.rank, level_pairs)
    if role is None:
        return seg
    if role[0] == "send":
        merge_send(tr, role[1], seg, views, prune_ratio, importance_fn=importance_fn)
        return None
    return merge_recv(tr, role[1], seg, views, prune_ratio, src_to_dst, importance_fn=importance_fn)["merged"]
