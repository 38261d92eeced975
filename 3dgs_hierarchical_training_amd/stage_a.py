"""Stage A of the hierarchical trainer, sharded: relative-pose fits of consecutive frame pairs.

Reference: `for fidx in range(1, seq_len): compute_relative_pose(fidx, fidx-1)`
(/root/reference/trainer/ht3dgs_trainer.py:697-698).  Each call trains a single-image 3DGS on frame f-1 (<= 1000
iterations, :274-305, :352-365) and then fits the SE(3) pose that makes it explain frame f (300 iterations,
:308-333, :367-378); nothing is shared between pairs, and the results land in `pose_dict` under
`rel_pose_{a}_to_{b}` (+ `rel_pose_{a}_to_{a}.5`, `rel_pose_{a}.5_to_{b}` in the interpolated-frame mode, :427-429).
That is ~70 % of all render calls of a scene and embarrassingly parallel:

  * pair p = (p, p+1) goes to rank p % world (round-robin keeps the ranks' loads equal when frames vary smoothly);
  * every rank writes its results into a [P, 3, 4, 4] table (slot 0 = a -> b, slots 1 / 2 = the two half steps, identity
    when unused), the only collective is ONE `all_gather` of the ranks' own rows (<= 192 B per pair);
  * every rank ends with the full `pose_dict`, which stage B needs replicated (leaf pose chains :739-741, merge pose
    chains :783-790).

`fit_pair` is the per-pair work on the MI355X rasterizer: single-image training with the fused train step, then Adam on
the six tangent numbers through the fused pose action (`points_transform`, pose.py) -- the means never leave the kernels,
and the pose update between two renders is one kernel (`gsr_pose_step`: dL/dM -> dL/d(delta) -> Adam -> next M).
"""
from typing import Callable, Dict, List, Optional

import torch
import torch.distributed as dist


def pairs_of_rank(n_frames: int, rank: int, world: int) -> List[int]:
    """Pair indices p (frames p -> p+1) owned by `rank`."""
    return list(range(rank, n_frames - 1, world))


def gather_pose_table(local: Dict[int, torch.Tensor], n_frames: int, rank: int, world: int, device, group=None) -> torch.Tensor:
    """local: {pair index -> [3,4,4]} for this rank's pairs.  Returns the full [P,3,4,4] table on every rank."""
    P = n_frames - 1
    rows = (P + world - 1) // world                   # every rank contributes the same number of rows (padding = identity)
    mine = torch.eye(4, dtype=torch.float32).repeat(rows, 3, 1, 1)
    for j, p in enumerate(pairs_of_rank(n_frames, rank, world)):
        mine[j] = local[p].detach().float().cpu()
    mine = mine.to(device).contiguous()
    if world == 1:
        return mine[:P]
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    table = torch.empty((P, 3, 4, 4), dtype=torch.float32, device=device)
    for r in range(world):
        idx = pairs_of_rank(n_frames, r, world)
        if idx:
            table[torch.tensor(idx, device=device)] = out[r][:len(idx)]
    return table


def pose_dict_from_table(table: torch.Tensor, vfi: bool = False) -> Dict[str, torch.Tensor]:
    """The reference's `pose_dict` keys (:378, :427-429)."""
    d = {}
    for p in range(table.shape[0]):
        d[f"rel_pose_{p}_to_{p + 1}"] = table[p, 0]
        if vfi:
            d[f"rel_pose_{p}_to_{p}.5"] = table[p, 1]
            d[f"rel_pose_{p}.5_to_{p + 1}"] = table[p, 2]
    return d


def run_stage_a(n_frames: int, fit_fn: Callable[[int], torch.Tensor], device, rank: Optional[int] = None,
                world: Optional[int] = None, group=None, vfi: bool = False, concurrency: int = 1,
                fit_device=None, batch_fn: Optional[Callable[[List[int]], Dict[int, torch.Tensor]]] = None, batch: int = 1) -> Dict[str, torch.Tensor]:
    """fit_fn(p) -> [4,4] (or [3,4,4]) relative pose(s) of pair p.  Every rank returns the complete pose_dict.

    concurrency > 1 (with fit_device = the GPU the fits run on): that many of this rank's pairs are fitted at the same time, each on
    a HIP stream and a host thread of its own.  A single-image model of ~130 k Gaussians does not fill an MI355X -- its sorts are
    latency chains of a few dozen workgroups, its blends run 17 waves per CU -- and two independent fits interleave on the device:
    0.338 -> 0.262 ms per iteration and pair (tools/two_streams_probe.py; three or four gain nothing more, the Python of the
    loops then takes turns on the interpreter lock)."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = pairs_of_rank(n_frames, rank, world)
    if batch_fn is not None and batch > 1 and len(mine) > 1:
        # `batch` of this rank's pairs per launch chain (fit_pairs_batched): the GPU sees one model `batch` times the size.
        # With concurrency > 1 two such chains run at a time, each on its own stream and host thread: one chain's sorts and scans
        # (latency chains, a few dozen workgroups) then overlap the other's blends.
        groups = [mine[lo:lo + batch] for lo in range(0, len(mine), batch)]
        results = {}
        if concurrency > 1 and fit_device is not None and torch.device(fit_device).type == "cuda" and len(groups) > 1:
            import threading
            from concurrent.futures import ThreadPoolExecutor
            fdev = torch.device(fit_device)
            main = torch.cuda.current_stream(fdev)
            tls = threading.local()

            def work_group(g):
                if not hasattr(tls, "stream"):
                    tls.stream = torch.cuda.Stream(fdev)
                    tls.stream.wait_stream(main)
                with torch.cuda.stream(tls.stream):
                    r = batch_fn(g)
                    tls.stream.synchronize()
                return r
            with ThreadPoolExecutor(max_workers=concurrency) as ex:
                for r in ex.map(work_group, groups):
                    results.update(r)
        else:
            for g in groups:
                results.update(batch_fn(g))
    elif concurrency > 1 and fit_device is not None and torch.device(fit_device).type == "cuda" and len(mine) > 1:
        import threading
        from concurrent.futures import ThreadPoolExecutor
        fdev = torch.device(fit_device)
        main = torch.cuda.current_stream(fdev)
        tls = threading.local()

        def work(p):
            if not hasattr(tls, "stream"):
                tls.stream = torch.cuda.Stream(fdev)
                tls.stream.wait_stream(main)          # what the main stream produced so far (targets, ...) is visible
            with torch.cuda.stream(tls.stream):
                r = fit_fn(p)
                tls.stream.synchronize()
            return p, r
        with ThreadPoolExecutor(max_workers=concurrency) as ex:
            results = dict(ex.map(work, mine))
    else:
        results = {p: fit_fn(p) for p in mine}
    local = {}
    for p in mine:
        r = results[p]
        if r.dim() == 2:
            eye = torch.eye(4, dtype=r.dtype, device=r.device)
            r = torch.stack((r, eye, eye))
        local[p] = r
    return pose_dict_from_table(gather_pose_table(local, n_frames, rank, world, device, group), vfi)


def fit_pair(seq, p: int, device, n_points: int = 100_000, single_image_iters: int = 1000, pose_iters: int = 300,
             pose_lr: float = 2e-3, seed: int = 0, init: str = "pixels", fused_pose_step: bool = True) -> torch.Tensor:
    """compute_relative_pose(p+1, p) on the HIP rasterizer (:336-380): returns rel_pose_{p}_to_{p+1} [4,4] (CPU).

    1. single-image 3DGS of frame p in that frame's own camera coordinates (identity pose), initialised from the frame's
       un-projected depth (sequence.pixel_scene; :352-363), fused train step, no densification (:352-365; early exit on
       PSNR > 35 after 500 iterations as :300-301);
    2. freeze the Gaussians, fit delta in Exp(delta) (tangent at the identity, `init_RT(None)` :370) on frame p+1 with
       Adam through `points_transform` (pose.py): loss = the same photometric loss."""
    from . import pose as pose_mod
    from . import train_step as ts
    from .loss import fused_photometric_loss
    from .rasterizer import rasterize_gaussians_raw
    if init == "pixels":     # one Gaussian per strided pixel of frame p, un-projected with its depth (about n_points of them)
        stride = max(1, int(round((seq.W * seq.H / max(1, n_points)) ** 0.5)))
        scene = seq.pixel_scene(p, stride=stride, seed=seed)
    else:                    # a perturbed subset of the ground-truth cloud
        scene = seq.leaf_scene(p, n_points, seed=seed + p)
    params = ts.GaussianParams(scene, device)
    # a stage-A model is a fresh HTGaussianModel: active SH degree 0 with 16 coefficients stored, and its <= 1 000 + 300 iterations
    # never reach an `oneupSHdegree` (gaussian_model_ht.py:68; train_single_image_3DGS / train_relative_pose have none)
    params.active_sh_degree = 0
    ident = ts.with_sh_degree(seq.settings_for_pose(torch.eye(4)), 0)
    tgt0, tgt1 = seq.target(p), seq.target(p + 1)
    for it in range(1, single_image_iters + 1):
        pkg = ts.train_step(params, ident, tgt0, next_settings=ident)      # same view every step: its preprocess rides in the backward
        if it > 500 and it % 50 == 0:
            with torch.no_grad():
                mse = ((pkg["raw_image"].clamp(0, 1) - tgt0) ** 2).mean()
            if float(-10 * torch.log10(mse.clamp_min(1e-12))) > 35:
                break
    raw = params.raw()
    m2d = torch.zeros_like(raw["_xyz"])
    if not fused_pose_step:      # the torch statement of the loop: exponential map + its autograd + torch.optim.Adam (~50 tiny kernels)
        delta = torch.zeros(6, device=device, requires_grad=True)
        opt = torch.optim.Adam([delta], lr=pose_lr)
        pose7 = torch.tensor([0, 0, 0, 0, 0, 0, 1.0], device=device)
        for _ in range(pose_iters):
            opt.zero_grad(set_to_none=True)
            M = pose_mod.retr_matrix(delta, pose7)
            img = rasterize_gaussians_raw(raw["_xyz"], m2d, raw["_features_dc"], raw["_features_rest"], raw["_opacity"],
                                          raw["_scaling"], raw["_rotation"], ident, points_transform=M)[0]
            fused_photometric_loss(img, tgt1, 0.2, clamp=True).backward()
            opt.step()
        return pose_mod.retr_matrix(delta.detach(), pose7).cpu()
    # fused: one one-thread kernel between two renders (gsr_pose_step) takes dL/dM, applies Adam to the six tangent numbers and
    # writes the next M where the next render reads it -- 1.9 -> 0.4 ms per iteration at 130 k Gaussians
    from . import _ext
    ops = _ext.load()
    delta = torch.zeros(6, device=device)
    m, v = torch.zeros(6, device=device), torch.zeros(6, device=device)
    none = torch.empty(0, device=device)
    M = torch.zeros(3, 4, device=device)
    ops.pose_step(delta, m, v, none, none, M, pose_lr, 0.9, 0.999, 1e-8, 0)              # M = Exp(0) = identity
    for it in range(1, pose_iters + 1):
        Mi = M.detach().requires_grad_(True)
        img = rasterize_gaussians_raw(raw["_xyz"], m2d, raw["_features_dc"], raw["_features_rest"], raw["_opacity"],
                                      raw["_scaling"], raw["_rotation"], ident, points_transform=Mi)[0]
        fused_photometric_loss(img, tgt1, 0.2, clamp=True).backward()
        ops.pose_step(delta, m, v, Mi.grad, none, M, pose_lr, 0.9, 0.999, 1e-8, it)       # torch.optim.Adam's defaults
    out = torch.eye(4)
    out[:3] = M.detach().cpu()
    return out


def fit_pairs_batched(seq, pairs: List[int], device, n_points: int = 100_000, single_image_iters: int = 1000, pose_iters: int = 300,
                      pose_lr: float = 2e-3, seed: int = 0) -> Dict[int, torch.Tensor]:
    """`fit_pair` for several pairs AT ONCE: the B single-image models live in one parameter store and every step of the B fits is
    ONE launch chain (batched.BatchedGaussianParams / include/gsr.h GsrBatch) -- B renders, B losses, B Adam updates and the B
    hand-overs to the next step per pass over the kernels, instead of B chains of ~17 kernels that each fill a fraction of the
    chip.  Per model the arithmetic is what `fit_pair` does (bit-identical renders and updates, tests/test_gpu_batched.py); what
    differs is the early exit of the image phase: the reference leaves a model's loop when its PSNR passes 35 dB after 500
    iterations (ht3dgs_trainer.py:300-301), a batch runs until ALL its models have passed (or the iteration cap) -- a model that
    is already there simply trains a little longer.  Returns {pair -> rel_pose [4,4] (CPU)}."""
    from . import batched as bt
    from . import train_step as ts
    from . import _ext
    from .loss import fused_photometric_loss
    from .rasterizer import rasterize_gaussians_raw
    B = len(pairs)
    if B == 1:
        return {pairs[0]: fit_pair(seq, pairs[0], device, n_points, single_image_iters, pose_iters, pose_lr, seed)}
    stride = max(1, int(round((seq.W * seq.H / max(1, n_points)) ** 0.5)))
    scenes = [seq.pixel_scene(p, stride=stride, seed=seed) for p in pairs]
    params = bt.BatchedGaussianParams(scenes, device)
    params.active_sh_degree = 0          # fresh models: degree 0, 16 coefficients stored (see fit_pair)
    ident1 = ts.with_sh_degree(seq.settings_for_pose(torch.eye(4)), 0)
    ident = bt.batch_settings([ident1] * B, device)
    tgt0 = torch.stack([seq.target(p) for p in pairs])
    tgt1 = torch.stack([seq.target(p + 1) for p in pairs])
    for it in range(1, single_image_iters + 1):
        pkg = ts.train_step(params, ident, tgt0, next_settings=ident)
        if it > 500 and it % 50 == 0:
            with torch.no_grad():
                mse = ((pkg["raw_image"].clamp(0, 1) - tgt0) ** 2).flatten(1).mean(dim=1)
            if float((-10 * torch.log10(mse.clamp_min(1e-12))).min()) > 35:
                break
    raw = params.raw()
    m2d = torch.zeros_like(raw["_xyz"])
    ops = _ext.load()
    delta = torch.zeros(B, 6, device=device)
    m, v = torch.zeros(B, 6, device=device), torch.zeros(B, 6, device=device)
    none = torch.empty(0, device=device)
    M = torch.zeros(B, 3, 4, device=device)
    for b in range(B):
        ops.pose_step(delta[b], m[b], v[b], none, none, M[b], pose_lr, 0.9, 0.999, 1e-8, 0)              # M_b = Exp(0) = identity
    for it in range(1, pose_iters + 1):
        Mi = M.detach().requires_grad_(True)
        img = rasterize_gaussians_raw(raw["_xyz"], m2d, raw["_features_dc"], raw["_features_rest"], raw["_opacity"], raw["_scaling"],
                                      raw["_rotation"], ident, points_transform=Mi, batch_first_block=params.first_block)[0]
        fused_photometric_loss(img, tgt1, 0.2, clamp=True).backward()          # sum of the B losses: every transform gets its own gradient
        g = Mi.grad
        for b in range(B):                                                        # one one-thread kernel per model (gsr_pose_step)
            ops.pose_step(delta[b], m[b], v[b], g[b], none, M[b], pose_lr, 0.9, 0.999, 1e-8, it)
    out = {}
    Mc = M.detach().cpu()
    for b, p in enumerate(pairs):
        T = torch.eye(4)
        T[:3] = Mc[b]
        out[p] = T
    return out
