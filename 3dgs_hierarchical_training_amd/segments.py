"""One-segment-per-GPU sharding of the hierarchical trainer and the merge-step exchange.

The reference trains 2^level leaf segments one after another on cuda:0 and merges neighbours pairwise
(/root/reference/trainer/ht3dgs_trainer.py:710-804; merge_two_3DGS :214-272).  Leaf segments are independent
(:729-753), so here rank r owns leaf segment r; the ONLY data exchange is at a merge, where the source
child travels src -> dst point-to-point:

  * the Gaussian tensors, 59 fp32 = 236 B per Gaussian (_xyz, _features_dc, _features_rest, _opacity, _scaling,
    _rotation -- :257-267), **un-pruned**: with the 'base' multi-source supervision the child is the frozen teacher
    of the parent's phase 1 (:757, :866-883), so the whole child is sent once and the importance mask (:247-253,
    computed on the child's home rank, in parallel with the destination's own) is applied at the destination;
  * the drop mask, one byte per Gaussian;
  * the child's frames and their poses (`to_visit_frames`, `start_fidx`, the pose list P -- :734-735, :766 and
    /root/reference/scene/gaussian_model_ht.py:363-377; 4x4 matrices here) and `global_iteration`.

xGMI is a full mesh of point-to-point links, so each pair of a merge level uses its own link; no ring collective
is involved.  Backend "nccl" is RCCL on ROCm; the same code runs over gloo on CPU for the world_size-2/4 tests, and
through `LocalTransport` when one process walks the whole tree on one device (the reference's own execution order).
"""
import time
from collections import deque
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

SEGMENT_KEYS = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
_TRAILING = {"_xyz": (3,), "_features_dc": (1, 3), "_features_rest": (15, 3), "_opacity": (1,), "_scaling": (3,),
             "_rotation": (4,)}
FLOATS_PER_GAUSSIAN = 59


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


# ---- transports -------------------------------------------------------------------------------------------------------
class DistTransport:
    """torch.distributed point-to-point (RCCL on the GPUs, gloo in the CPU tests).

    host_staging=True moves device tensors through host memory around the send / receive: gloo has no device-side
    point-to-point, and with it the SAME runner code can be driven by several processes that share ONE GPU
    (run_segments.py --backend gloo --one-device) -- the way the multi-process walk of the merge tree is exercised on a
    one-GPU box.  With RCCL the tensors go device to device and this stays off.

    A message a rank addresses to ITSELF (a rank that holds both children of a merge; the one-GPU test of the RCCL
    point-to-point path) cannot use the blocking pair -- `send` would wait for a receive that is never posted.  It is held
    back until the matching `recv`, and both halves then go out as ONE grouped call (`batch_isend_irecv([isend, irecv])`),
    which RCCL accepts for a self pair (tools/rccl_selfpair.py on MI355X, RCCL 2.26.6: 1 MiB and 64 MiB device tensors,
    payload intact)."""

    def __init__(self, group=None, host_staging: bool = False):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.host_staging = host_staging
        self._to_self = deque()
        self._self_over_wire = dist.get_backend(group) == "nccl"

    def send(self, t: torch.Tensor, dst: int):
        if self.host_staging and t.is_cuda:
            t = t.detach().cpu()
        if dst == self.rank:
            # a copy, not an alias: a blocking send lets the caller reuse the buffer as soon as send() returns
            self._to_self.append(t.detach().clone())
            return
        dist.send(t, dst, group=self.group)

    def recv(self, t: torch.Tensor, src: int):
        buf = torch.empty(t.shape, dtype=t.dtype, device="cpu") if (self.host_staging and t.is_cuda) else t
        if src == self.rank:
            if not self._to_self:
                raise RuntimeError(f"DistTransport: rank {self.rank} receives from itself before it sent")
            m = self._to_self.popleft()
            if m.shape != buf.shape or m.dtype != buf.dtype:
                raise RuntimeError(f"DistTransport: self message {tuple(m.shape)} {m.dtype} does not match the receive buffer "
                                   f"{tuple(buf.shape)} {buf.dtype}")
            if self._self_over_wire:
                for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, m, self.rank, group=self.group),
                                                 dist.P2POp(dist.irecv, buf, self.rank, group=self.group)]):
                    w.wait()
            else:           # gloo has no pair from a rank to itself ("Pair is not connected"): the mailbox alone
                buf.copy_(m)
        else:
            dist.recv(buf, src, group=self.group)
        if buf is not t:
            t.copy_(buf)


class LocalTransport:
    """In-process mailboxes: one process plays every rank in turn on one device (senders of a level run before its
    receivers).  A send is a device copy, so the 'wire' time it reports is an HBM copy, not a link."""

    def __init__(self, world: int):
        self.world = world
        self.rank = 0           # set by the driver before each virtual rank acts
        self.box: Dict[Tuple[int, int], deque] = {}

    def send(self, t: torch.Tensor, dst: int):
        self.box.setdefault((self.rank, dst), deque()).append(t.detach().clone())

    def recv(self, t: torch.Tensor, src: int):
        q = self.box.get((src, self.rank))
        if not q:
            raise RuntimeError(f"LocalTransport: rank {self.rank} receives from {src} before it sent")
        m = q.popleft()
        if m.shape != t.shape or m.dtype != t.dtype:
            raise RuntimeError(f"LocalTransport: message {tuple(m.shape)} {m.dtype} does not match the receive buffer "
                               f"{tuple(t.shape)} {t.dtype}")
        t.copy_(m)


def _sync(t: torch.Tensor):
    if t.is_cuda:
        torch.cuda.synchronize(t.device)


def link_selftest(tr, schedule: List[List[Tuple[int, int]]], device, nbytes: int = 1 << 20) -> Dict:
    """Before the tree is walked: every merge pair of every level exchanges `nbytes` of a rank-dependent pattern in the direction
    the merge will use (src -> dst) and echoes it back; both ends check what arrived.  Returns {'ok', 'pairs': [...], 'ms'} for
    this rank; a mismatch names the pair (run_segments aborts with a JSON error line instead of training for minutes towards a
    broken exchange).  Levels run one after the other, the pairs of a level concurrently -- the order of the merges themselves."""
    import time
    t0 = time.perf_counter()
    n = nbytes // 4
    idx = torch.arange(n, dtype=torch.int32, device=device)
    pairs, ok = [], True
    for lv, level in enumerate(schedule):
        role = partner(tr.rank, level)
        if role is None:
            continue
        kind, peer = role
        src, dst = (tr.rank, peer) if kind == "send" else (peer, tr.rank)
        want = idx * 31 + 7 * src + lv                  # what travels src -> dst
        echo = want ^ 0x5a5a5a5a                          # and back
        buf = torch.empty_like(idx)
        if kind == "send":
            tr.send(want, peer)
            tr.recv(buf, peer)
            good = bool(torch.equal(buf, echo))
        else:
            tr.recv(buf, peer)
            good = bool(torch.equal(buf, want))
            tr.send(buf ^ 0x5a5a5a5a, peer)
        ok = ok and good
        pairs.append({"level": lv, "src": src, "dst": dst, "ok": good})
    _sync(idx)
    return {"ok": ok, "pairs": pairs, "bytes": n * 4, "ms": 1e3 * (time.perf_counter() - t0)}


# ---- the child message -----------------------------------------------------------------------------------------------
def pack_segment(seg: Dict[str, torch.Tensor]) -> torch.Tensor:
    n = seg["_xyz"].shape[0]
    return torch.cat([seg[key].detach().reshape(n, -1).float() for key in SEGMENT_KEYS], dim=1).contiguous()


def unpack_segment(flat: torch.Tensor) -> Dict[str, torch.Tensor]:
    n = flat.shape[0]
    seg, o = {}, 0
    for key in SEGMENT_KEYS:
        w = 1
        for d in _TRAILING[key]:
            w *= d
        seg[key] = flat[:, o:o + w].reshape((n,) + _TRAILING[key]).contiguous()
        o += w
    return seg


def send_child(tr, dst: int, seg: Dict[str, torch.Tensor], drop: Optional[torch.Tensor] = None,
               frames: Optional[List[int]] = None, poses: Optional[torch.Tensor] = None, start_fidx: int = 0,
               global_iteration: int = 0, sh_degree: int = -1) -> Dict:
    """Header (sizes, variable N, the child's ACTIVE SH degree), then the un-pruned 59-float rows, then mask / frames / poses.
    Returns {'bytes', 'ms'} (ms includes the wait for the receiver).  sh_degree: the degree the child was trained at -- the receiver
    renders it as a teacher at THAT degree (uneven frame splits or iteration counts can leave the two sides of a merge on
    different sides of an `oneupSHdegree` boundary; -1 = not stated)."""
    dev = seg["_xyz"].device
    n = seg["_xyz"].shape[0]
    frames = list(frames or [])
    t0 = time.perf_counter()
    hdr = torch.tensor([n, len(frames), int(start_fidx), int(global_iteration), 0 if drop is None else 1, int(sh_degree)],
                       dtype=torch.int64, device=dev)
    tr.send(hdr, dst)
    nbytes = hdr.numel() * 8
    flat = pack_segment(seg)
    tr.send(flat, dst)
    nbytes += flat.numel() * 4
    if drop is not None:
        m = drop.to(device=dev, dtype=torch.uint8).contiguous()
        tr.send(m, dst)
        nbytes += m.numel()
    if frames:
        tr.send(torch.tensor(frames, dtype=torch.int64, device=dev), dst)
        p = poses.detach().to(device=dev, dtype=torch.float32).reshape(len(frames), 16).contiguous()
        tr.send(p, dst)
        nbytes += len(frames) * (8 + 64)
    _sync(flat)
    return {"bytes": nbytes, "ms": 1e3 * (time.perf_counter() - t0)}


def recv_child(tr, src: int, device) -> Dict:
    """Counterpart of send_child: {'seg', 'drop', 'frames', 'poses', 'start_fidx', 'global_iteration', 'sh_degree', 'bytes', 'ms'}."""
    t0 = time.perf_counter()
    hdr = torch.zeros(6, dtype=torch.int64, device=device)
    tr.recv(hdr, src)
    n, nf, start_fidx, giter, has_mask, sh_degree = (int(v) for v in hdr.tolist())
    nbytes = 48
    flat = torch.empty((n, FLOATS_PER_GAUSSIAN), dtype=torch.float32, device=device)
    tr.recv(flat, src)
    nbytes += flat.numel() * 4
    drop = None
    if has_mask:
        m = torch.empty(n, dtype=torch.uint8, device=device)
        tr.recv(m, src)
        drop = m.bool()
        nbytes += n
    frames, poses = [], None
    if nf:
        f = torch.empty(nf, dtype=torch.int64, device=device)
        tr.recv(f, src)
        frames = [int(v) for v in f.tolist()]
        poses = torch.empty((nf, 16), dtype=torch.float32, device=device)
        tr.recv(poses, src)
        poses = poses.reshape(nf, 4, 4)
        nbytes += nf * (8 + 64)
    _sync(flat)
    return {"seg": unpack_segment(flat), "drop": drop, "frames": frames, "poses": poses, "start_fidx": start_fidx,
            "global_iteration": giter, "sh_degree": sh_degree, "bytes": nbytes, "ms": 1e3 * (time.perf_counter() - t0)}


# ---- round-1 names (kept: the world_size-2 exchange test and external callers use them) ---------------------------------
def send_segment(seg: Dict[str, torch.Tensor], dst: int, extra: Optional[torch.Tensor] = None, group=None) -> None:
    """Count first (N differs per segment), then one flat 59-float-per-Gaussian message (+ an optional float vector)."""
    n = seg["_xyz"].shape[0]
    dev = seg["_xyz"].device
    k = 0 if extra is None else extra.numel()
    dist.send(torch.tensor([n, k], dtype=torch.int64, device=dev), dst, group=group)
    dist.send(pack_segment(seg), dst, group=group)
    if k:
        dist.send(extra.detach().float().contiguous().reshape(-1), dst, group=group)


def recv_segment(src: int, device, group=None) -> Tuple[Dict[str, torch.Tensor], Optional[torch.Tensor]]:
    hdr = torch.zeros(2, dtype=torch.int64, device=device)
    dist.recv(hdr, src, group=group)
    n, k = int(hdr[0].item()), int(hdr[1].item())
    flat = torch.empty((n, FLOATS_PER_GAUSSIAN), dtype=torch.float32, device=device)
    dist.recv(flat, src, group=group)
    extra = None
    if k:
        extra = torch.empty(k, dtype=torch.float32, device=device)
        dist.recv(extra, src, group=group)
    return unpack_segment(flat), extra


def merge_segments(dst_seg: Dict[str, torch.Tensor], src_seg: Dict[str, torch.Tensor], dst_keep: torch.Tensor,
                   src_keep: torch.Tensor, src_to_dst: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """merge_two_3DGS (:233-271): prune both by their importance masks, move the source points by the 4x4
    relative transform (homogeneous, with the divide of :252-254), append."""
    out = {}
    xyz = src_seg["_xyz"]
    if src_to_dst is not None:
        T = src_to_dst.to(xyz)
        h = xyz @ T[:3, :3].t() + T[:3, 3]
        w = xyz @ T[3, :3] + T[3, 3]
        xyz = h / w.unsqueeze(1)
    xyz = xyz[src_keep]
    for key in SEGMENT_KEYS:
        s = xyz if key == "_xyz" else src_seg[key][src_keep]
        out[key] = torch.cat([dst_seg[key][dst_keep], s], dim=0)
    return out
