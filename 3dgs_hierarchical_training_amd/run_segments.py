#!/usr/bin/env python3
"""BASELINE config 4: hierarchical training with one leaf segment per GPU and a point-to-point exchange at each merge.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
         3dgs_hierarchical_training_amd/run_segments.py --frames 40 --leaf-gaussians 200000
  python 3dgs_hierarchical_training_amd/run_segments.py --local --segments 8      # whole tree on one GPU, in order

Stage B of `HTGaussianTrainer.hierarchical_training` (/root/reference/trainer/ht3dgs_trainer.py:710-813) re-cut for
eight GPUs:

  reference (one GPU, sequential)                       here (rank r = leaf r)
  ------------------------------------------------      ---------------------------------------------------------------
  for each leaf: init_leaf_3DGS + train_leaf_3DGS       every rank trains its own leaf at the same time, no communication
    :729-753
  after every second segment: merge_two_3DGS            level k: ranks (2^(k+1) j, 2^(k+1) j + 2^k) pair up; both compute
    (importance of both, prune both, move the             their own child's importance at the same time, the source ships
    source by inverse(get_RT(src.start_fidx)),            the UN-PRUNED child + drop mask + frames + poses over its own
    append) :767-780, :214-272                            xGMI link, the destination applies the masks, moves, appends
  poses of the source's frames chained on :783-790      same, from the (replicated) relative-pose table of stage A
  train_nonleaf_3DGS_phase1 with the two frozen         same, on the destination rank; the teachers are the two un-pruned
    children as teachers for virtual views :757,          children it now holds (its own snapshot + the received one)
    :815-900
  train_nonleaf_3DGS_phase2 on the real frames :762     same

Source ranks are idle after their send (tree reduction: 8 -> 4 -> 2 -> 1 busy GPUs).  Frames, targets and relative
poses are synthetic (sequence.py); the relative-pose table stands for stage A's result (stage_a.py shards that stage).

`RankRunner` holds one rank's state; `run_local` plays all ranks in turn on one device through `LocalTransport`
(senders of a level before its receivers) -- the same code path, and the reference's own execution order.
Every rank prints one JSON line per phase: leaf, and per level {importance_ms, send_ms | recv_ms, bytes, ...}.
"""
import argparse
import importlib
import json
import os
import random
import sys
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

if __package__ in (None, ""):      # executed as a script: make the oddly named package importable
    _root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if _root not in sys.path:
        sys.path.insert(0, _root)
    _pkg = importlib.import_module("3dgs_hierarchical_training_amd")
    hierarchy = importlib.import_module("3dgs_hierarchical_training_amd.hierarchy")
    segments = importlib.import_module("3dgs_hierarchical_training_amd.segments")
    sequence = importlib.import_module("3dgs_hierarchical_training_amd.sequence")
    ts = importlib.import_module("3dgs_hierarchical_training_amd.train_step")
    densify = importlib.import_module("3dgs_hierarchical_training_amd.densify")
    pose_mod = importlib.import_module("3dgs_hierarchical_training_amd.pose")
    host_mod = importlib.import_module("3dgs_hierarchical_training_amd.host")
    stage_a = importlib.import_module("3dgs_hierarchical_training_amd.stage_a")
else:
    from . import densify, hierarchy, segments, sequence
    from . import host as host_mod
    from . import stage_a
    from . import pose as pose_mod
    from . import train_step as ts


def emit_line(rec):
    """One report record = one line = ONE write (several ranks share the launcher's stdout: print() issues the text and the
    newline separately and the lines of two ranks can interleave)."""
    sys.stdout.write(json.dumps(rec) + "\n")
    sys.stdout.flush()


@dataclass
class HTConfig:
    """Iteration counts follow /root/reference/arguments/full/Tanks/Ballroom.yml:5-16 in meaning; the defaults are
    shortened so the harness finishes in minutes."""
    frames: int = 40
    width: int = 980
    height: int = 545
    sh_degree: int = 3                          # max_sh_degree: 16 stored coefficients
    start_sh_degree: int = 0                    # active degree of a new leaf (gaussian_model_ht.py:68) ...
    sh_up_every: int = 1000                     # ... raised after every this many global iterations (ht3dgs_trainer.py:580-581)
    gt_gaussians: int = 400_000
    leaf_gaussians: int = 200_000
    leaf_iters_per_frame: int = 30              # single_step
    phase1_iters_per_frame: int = 5             # mss_phase1_iteration_per_frame (50 in Ballroom.yml)
    phase1_ratio: float = 0.5                   # mss_phase1_ratio
    phase2_iters_per_frame: List[int] = field(default_factory=lambda: [10, 10, 10])   # num_iterations_per_frame_each_level
    prune_ratio: float = 0.5                    # Ballroom.yml:44
    importance_views: int = 0                   # 0 = all of the child's frames (the reference), k = every (len/k)-th
    densify: bool = False
    seed: int = 0
    optimizer: str = "hip"
    fused: bool = True
    stage_a_concurrency: int = 2                # frame pairs fitted at the same time per GPU (streams + host threads), stage_a.run_stage_a
    stage_a_batch: int = 0                      # frame pairs fitted in ONE launch chain per GPU (stage_a.fit_pairs_batched, GsrBatch); 1 = off;
                                                # 0 = automatic: 4 per chain with two chains at a time when the host has the CPUs for two launching
                                                # threads (39 pairs: 6.3 s), 8 per chain otherwise (6.9 s); round 2's one pair per chain on two streams: 8.1 s
    fit_pose: bool = False                      # refine each frame's pose while training on it (training_setup(fit_pose=True), :733)
    pose_lr: float = 1e-5                       # Adam with eps 1e-15 moves a pose by ~lr per step whatever the gradient: on these
                                                # frames (1-2 px of motion per frame) 5e-6..2e-5 gains 0.4-0.6 dB over fixed stage-A
                                                # poses, 1e-4 loses 0.5 dB (tools/pose_lr_sweep.sh); the reference's rotation_lr is 1e-3


class Segment:
    """One 3DGS model and what the trainer keeps next to it (`gs_render.start_fidx / to_visit_frames /
    global_iteration`, :733-735, :766; `get_RT`, gaussian_model_ht.py:150-160)."""

    def __init__(self, params, frames: List[int], start_fidx: int, poses: Dict[int, torch.Tensor], global_iteration: int = 0):
        self.params, self.frames, self.start_fidx, self.poses, self.global_iteration = params, list(frames), start_fidx, dict(poses), global_iteration
        self.densifier = None
        self.pose_states = {}        # frame -> train_step.PoseState, while its pose is being refined (cfg.fit_pose)

    def pose_state(self, f: int, template, device, lr: float):
        ps = self.pose_states.get(f)
        if ps is None:
            ps = self.pose_states[f] = ts.CameraPoseState(template, self.poses[f], device, lr=lr)
            if f == self.start_fidx:
                ps.freeze()          # the segment's start frame fixes the gauge
        return ps

    def sync_poses(self):
        """Refined transforms back into the pose table (what merges, importance views and evaluation read).  The pose states --
        tangent numbers, Adam moments, step counts -- live as long as the segment does, as the reference's per-frame
        `camera_optimizer[fidx]` lives from one `training_setup` to the next (gaussian_model_ht.py:296-311); a merge builds a new
        Segment and with it fresh states."""
        for f, ps in self.pose_states.items():
            if ps.steps or f not in self.poses:
                self.poses[f] = ps.matrix()

    def pose_tensor(self) -> torch.Tensor:
        return torch.stack([self.poses[f] for f in self.frames])


class RankRunner:
    def __init__(self, rank: int, world: int, transport, seq, cfg: HTConfig, device, log=None, importance_fn=None,
                 step_fn=None, teacher_render_fn=None):
        """importance_fn / step_fn / teacher_render_fn replace the three places that reach the HIP rasterizer; the
        GPU-less gloo tests pass stand-ins so that the tree walk, the messages and the bookkeeping run on CPU."""
        self.rank, self.world, self.tr, self.seq, self.cfg, self.dev = rank, world, transport, seq, cfg, device
        self.importance_fn = importance_fn or hierarchy.calc_importance
        self.step_fn = step_fn
        self.teacher_render_fn = teacher_render_fn or hierarchy.render_raw
        self.level = world.bit_length() - 1
        self.parts = sequence.partition(cfg.frames, self.level)
        self.schedule = segments.merge_schedule(world)
        self.rng = random.Random(cfg.seed * 1000 + rank)
        self.seg: Optional[Segment] = None
        self.teachers = None
        self.log = log if log is not None else emit_line
        self.report = []

    def _emit(self, rec):
        rec = {"rank": self.rank, **rec}
        self.report.append(rec)
        self.log(rec)

    # ---- helpers ---------------------------------------------------------------------------------------------------
    def _settings(self, seg: Segment, f: int):
        """Raster settings of frame f under the segment's current pose of it -- built once per (frame, pose): a draw of a frame
        that was drawn before reuses the device tensors (building them is CPU matrix work + four host-to-device copies, a
        quarter of a leaf's wall time when done per draw)."""
        cache = seg.__dict__.setdefault("_settings_cache", {})
        pose = seg.poses[f]
        hit = cache.get(f)
        if hit is not None and hit[0] is pose:
            return hit[1]
        st = self.seq.settings_for_pose(pose)
        cache[f] = (pose, st)
        return st

    def _importance_views(self, seg: Segment):
        fr = seg.frames
        k = self.cfg.importance_views
        if k and len(fr) > k:
            fr = fr[::max(1, len(fr) // k)][:k]
        return [ts.with_sh_degree(self._settings(seg, f), seg.params.active_sh_degree) if self.step_fn is None else self._settings(seg, f)
                for f in fr]

    def _step(self, seg: Segment, settings, target, next_settings=None):
        """next_settings: the camera of the NEXT step when it is already drawn -- its preprocess then rides in this step's
        backward (train_step / "prepare in backward")."""
        seg.global_iteration += 1
        if self.step_fn is not None:
            return self.step_fn(seg, settings, target)
        up = self._sh_up(seg)
        ts.train_step(seg.params, settings, target, fused_optimizer=self.cfg.fused, densifier=seg.densifier,
                      iteration=seg.global_iteration, next_settings=next_settings,
                      next_sh_degree=min(seg.params.active_sh_degree + 1, seg.params.max_sh_degree) if up else None)
        if up:
            seg.params.oneup_sh_degree()

    def _sh_up(self, seg: Segment) -> bool:
        """`if self.global_iteration % 1000 == 0: oneupSHdegree()` after the step (ht3dgs_trainer.py:580-581, :636-637, :909-910)."""
        return self.cfg.sh_up_every > 0 and seg.global_iteration % self.cfg.sh_up_every == 0

    def _steps_over(self, seg: Segment, frames_drawn):
        """Train on a pre-drawn list of frames (the frame of step k + 1 is known at step k)."""
        if self.cfg.fit_pose and self.step_fn is None:
            # pose refinement: frame f's camera tensors are functions of Exp(delta_f) * pose_f; after each render of frame f its six
            # tangent numbers take one Adam step and the camera is rewritten in place (one kernel, train_step.CameraPoseState).
            # The segment's start frame fixes the gauge and is not refined.
            states = [seg.pose_state(v, self._settings(seg, v), self.dev, self.cfg.pose_lr) for v in frames_drawn]
            for k, v in enumerate(frames_drawn):
                seg.global_iteration += 1
                up = self._sh_up(seg)
                ts.train_step(seg.params, states[k].settings, self.seq.target(v), fused_optimizer=self.cfg.fused, densifier=seg.densifier,
                              iteration=seg.global_iteration, pose=states[k], next_pose=states[k + 1] if k + 1 < len(states) else None,
                              next_sh_degree=min(seg.params.active_sh_degree + 1, seg.params.max_sh_degree) if up else None)
                if up:
                    seg.params.oneup_sh_degree()
            seg.sync_poses()
            return
        st = [self._settings(seg, v) for v in frames_drawn]
        for k, v in enumerate(frames_drawn):
            self._step(seg, st[k], self.seq.target(v), st[k + 1] if k + 1 < len(st) else None)

    def _new_densifier(self, seg: Segment):
        if self.cfg.densify:
            seg.densifier = densify.Densifier(seg.params, scene_extent=5.0, cfg=densify.DensifyConfig(
                densify_from_iter=50, densification_interval=100, opacity_reset_interval=3000,
                max_points=4 * seg.params.num_points), seed=self.rank)

    # ---- leaf (:729-753) ---------------------------------------------------------------------------------------------
    def train_leaf(self):
        cfg = self.cfg
        frames = self.parts[self.level][self.rank]
        start = frames[0]
        scene = self.seq.leaf_scene(start, cfg.leaf_gaussians, seed=cfg.seed + 17 * self.rank)
        params = ts.GaussianParams(scene, self.dev, optimizer=cfg.optimizer)
        params.active_sh_degree = min(cfg.start_sh_degree, params.max_sh_degree)      # a new model starts at degree 0 (gaussian_model_ht.py:68)
        seg = Segment(params, frames, start, {start: torch.eye(4)})
        self._new_densifier(seg)
        t0 = time.perf_counter()
        visited, steps = [start], 0
        for f in frames[1:]:
            seg.poses[f] = self.seq.rel_pose(f - 1, f) @ seg.poses[f - 1]          # :739-741
            visited.append(f)
            drawn = [self.rng.choice(visited) for _ in range(cfg.leaf_iters_per_frame)]   # sample_a_training_frame, :482-505
            self._steps_over(seg, drawn)
            steps += len(drawn)
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)
        self.seg = seg
        self._emit({"phase": "leaf", "frames": [frames[0], frames[-1]], "steps": steps, "gaussians": params.num_points,
                    "ms": 1e3 * (time.perf_counter() - t0)})

    # ---- merge (:767-793, :214-272) ----------------------------------------------------------------------------------
    def role(self, k: int):
        return segments.partner(self.rank, self.schedule[k])

    def merge_send(self, k: int):
        seg, self.tr.rank = self.seg, self.rank
        st = hierarchy.merge_send(self.tr, self.role(k)[1], seg.params.raw(), self._importance_views(seg), self.cfg.prune_ratio,
                                  frames=seg.frames, poses=seg.pose_tensor(), start_fidx=seg.start_fidx,
                                  global_iteration=seg.global_iteration, importance_fn=self.importance_fn,
                                  sh_degree=seg.params.active_sh_degree)
        self.seg = None          # this GPU is free from here on
        self._emit({"phase": "merge", "level": k, **st})

    def merge_recv(self, k: int):
        seg, self.tr.rank = self.seg, self.rank
        # child coordinates -> this model's coordinates: inverse of get_RT(child.start_fidx), :778-780.  The child's
        # start frame is one of this model's own frames (the partitions overlap by two frames).
        src_to_dst = lambda msg: torch.linalg.inv(seg.poses[msg["start_fidx"]])
        out = hierarchy.merge_recv(self.tr, self.role(k)[1], seg.params.raw(), self._importance_views(seg),
                                   self.cfg.prune_ratio, src_to_dst, importance_fn=self.importance_fn)
        child = out["child"]
        # teachers render at the active degree their model was trained at: the destination's own, and the child's as its message
        # states it (uneven frame splits or iteration counts can leave the two on different sides of an oneupSHdegree boundary)
        tdeg = seg.params.active_sh_degree
        cdeg = child.get("sh_degree", -1)
        cdeg = tdeg if cdeg is None or cdeg < 0 else min(int(cdeg), seg.params.max_sh_degree)
        own_teacher = {"seg": {kk: v.clone() for kk, v in out["teachers"][0].items()}, "start_fidx": seg.start_fidx,
                       "frames": list(seg.frames), "sh_degree": tdeg}
        child_teacher = {"seg": out["teachers"][1], "start_fidx": child["start_fidx"], "frames": list(child["frames"]), "sh_degree": cdeg}
        self.teachers = [own_teacher, child_teacher]
        for f in child["frames"]:                                                    # :783-790
            if f not in seg.poses:
                seg.poses[f] = self.seq.rel_pose(f - 1, f) @ seg.poses[f - 1]
        frames = sorted(set(seg.frames + child["frames"]))                           # :796
        # (the merged model keeps the destination's active degree: `restore` carries it, gaussian_model_ht.py:107-124)
        params = ts.GaussianParams.from_raw(out["merged"], self.dev, sh_degree=seg.params.active_sh_degree, optimizer=self.cfg.optimizer)
        self.seg = Segment(params, frames, seg.start_fidx, seg.poses, global_iteration=0)   # :792-793
        self._new_densifier(self.seg)
        self._emit({"phase": "merge", "level": k, "sh_degree_dst": tdeg, "sh_degree_child": cdeg, "sh_degree_merged": tdeg,
                    **{kk: v for kk, v in out.items() if kk not in ("merged", "teachers", "child")}})

    # ---- non-leaf training (:757-764, :815-900) ------------------------------------------------------------------------
    def train_nonleaf(self, k: int):
        cfg, seg = self.cfg, self.seg
        frames = seg.frames
        t0 = time.perf_counter()
        n1 = cfg.phase1_iters_per_frame * len(frames)
        virtual = 0
        for _ in range(n1):
            f = self.rng.choice(frames)
            if self.rng.random() < cfg.phase1_ratio:
                alpha = self.rng.random()
                if f == frames[-1]:
                    f -= 1
                if f + 1 not in seg.poses:
                    continue
                p = pose_mod.interpolate_pose(seg.poses[f], seg.poses[f + 1], alpha)     # get_virtual_view, :462-479
                teacher = next((t for t in self.teachers[::-1] if f >= t["start_fidx"] and f in t["frames"]), None)
                if teacher is None:
                    raise ValueError(f"frame {f} belongs to no child")
                p_wrt_teacher = p @ torch.linalg.inv(seg.poses[teacher["start_fidx"]])    # :874
                pseudo = self.teacher_render_fn(teacher["seg"], ts.with_sh_degree(self.seq.settings_for_pose(p_wrt_teacher), teacher["sh_degree"])
                                                if self.step_fn is None else self.seq.settings_for_pose(p_wrt_teacher))
                self._step(seg, self.seq.settings_for_pose(p), pseudo)
                virtual += 1
            else:
                self._step(seg, self._settings(seg, f), self.seq.target(f))
        self.teachers = None                                                               # :758-760
        lvl = self.level - 1 - k
        per = cfg.phase2_iters_per_frame[min(lvl, len(cfg.phase2_iters_per_frame) - 1)]
        n2 = per * len(frames)
        self._steps_over(seg, [self.rng.choice(frames) for _ in range(n2)])
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)
        self._emit({"phase": "nonleaf", "level": k, "phase1_steps": n1, "virtual_views": virtual, "phase2_steps": n2,
                    "gaussians": seg.params.num_points, "ms": 1e3 * (time.perf_counter() - t0)})

    def evaluate(self) -> float:
        """Mean PSNR of the final model over its frames."""
        seg, tot = self.seg, 0.0
        with torch.no_grad():
            for f in seg.frames:
                img = ts.render(seg.params, self._settings(seg, f))["image"]
                mse = ((img - self.seq.target(f)) ** 2).mean().clamp_min(1e-12)
                tot += float(-10.0 * torch.log10(mse))
        return tot / len(seg.frames)

    # ---- one rank's whole run (distributed) ----------------------------------------------------------------------------
    def link_selftest(self, all_ok=None) -> bool:
        """1 MB along every edge of the merge tree this rank takes part in, before any training (segments.link_selftest).
        all_ok: callable(bool) -> bool that combines the ranks' verdicts (an all-reduce); every rank then leaves together."""
        self.tr.rank = self.rank
        st = segments.link_selftest(self.tr, self.schedule, self.dev)
        good = st["ok"] if all_ok is None else all_ok(st["ok"])
        self._emit({"phase": "link_selftest", **st, "all_ranks_ok": good})
        if not good:
            self._emit({"phase": "error", "error": "merge-tree link self-test failed", "pairs": [p for p in st["pairs"] if not p["ok"]]})
        return good

    def run(self, barrier=None, all_ok=None):
        # a rank whose link self-test failed must not leave alone while its peers walk on to a merge that then never completes: with
        # more than one rank the verdicts have to be combined
        if all_ok is None and self.world > 1 and not isinstance(self.tr, segments.LocalTransport):
            raise RuntimeError("RankRunner.run: with world > 1 pass all_ok (a MIN all-reduce of the ranks' link self-test verdicts)")
        if not self.link_selftest(all_ok):
            raise SystemExit(3)
        self.train_leaf()
        for k in range(len(self.schedule)):
            if barrier is not None:
                barrier()
            if self.seg is None:
                continue
            r = self.role(k)
            if r is None:
                continue
            if r[0] == "send":
                self.merge_send(k)
            else:
                self.merge_recv(k)
                self.train_nonleaf(k)
        return self.seg


def run_local(world: int, seq, cfg: HTConfig, device, log=None):
    """All ranks in turn on one device (the reference's execution order).  Returns (root RankRunner, report)."""
    tr = segments.LocalTransport(world)
    runners = [RankRunner(r, world, tr, seq, cfg, device, log=log) for r in range(world)]
    for rr in runners:
        rr.train_leaf()
    for k in range(len(runners[0].schedule)):
        pairs = runners[0].schedule[k]
        for dst, src in pairs:
            runners[src].merge_send(k)
        for dst, src in pairs:
            runners[dst].merge_recv(k)
            runners[dst].train_nonleaf(k)
    return runners[0], [rec for rr in runners for rec in rr.report]


def run_stage_a_on(seq, cfg, dev, spec, rank: int, world: int, group=None, log=None, gather_device=None):
    """Stage A of hierarchical_training (ht3dgs_trainer.py:697-698) on this rank's share of the frame pairs; every rank ends
    with the full pose table and adopts it.  Returns the report record."""
    n_points, image_iters, pose_iters = spec
    t0 = time.perf_counter()
    mine = stage_a.pairs_of_rank(cfg.frames, rank, world)
    for f in sorted({q for p in mine for q in (p, p + 1)}):      # targets / depths rendered once, on the main stream, before the workers start
        seq.target(f)
    for p in mine:
        seq.depth(p)
    # concurrent fits per GPU, as far as the host allows: every fitting thread keeps about one CPU busy (launching, polling for the
    # instance count), and the ranks of a node share the container's CPU quota
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "1" if world == 1 else str(world)))
    conc = max(1, min(cfg.stage_a_concurrency, host_mod.usable_cpus()[0] // (2 * max(1, local_world))))
    batch = cfg.stage_a_batch if cfg.stage_a_batch > 0 else (4 if conc > 1 else 8)
    # the kernels' limits on a batch (include/gsr.h GsrBatch): 16 models per launch chain, B * tile rows <= 4095
    tiles_y = (cfg.height + 15) // 16
    batch = max(1, min(batch, 16, 4095 // max(1, tiles_y)))
    table = stage_a.run_stage_a(cfg.frames, lambda p: stage_a.fit_pair(seq, p, dev, n_points=n_points, single_image_iters=image_iters,
                                                                        pose_iters=pose_iters, seed=cfg.seed),
                                gather_device or dev, rank=rank, world=world, group=group, concurrency=conc, fit_device=dev,
                                batch=batch,
                                batch_fn=lambda ps: stage_a.fit_pairs_batched(seq, ps, dev, n_points=n_points, single_image_iters=image_iters,
                                                                              pose_iters=pose_iters, seed=cfg.seed))
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    err = max(float((table[f"rel_pose_{p}_to_{p + 1}"].cpu() - seq.true_rel_pose(p, p + 1)).abs().max()) for p in range(cfg.frames - 1))
    ident = max(float((torch.eye(4) - seq.true_rel_pose(p, p + 1)).abs().max()) for p in range(cfg.frames - 1))
    seq.use_pose_table(table)
    rec = {"rank": rank, "phase": "stage_a", "pairs_total": cfg.frames - 1, "pairs_here": len(stage_a.pairs_of_rank(cfg.frames, rank, world)),
           "gaussians": n_points, "image_iters": image_iters, "pose_iters": pose_iters,
           "pairs_at_a_time": batch * conc if batch > 1 else conc,
           "mode": f"batched: {batch} pairs per launch chain, {conc} chain(s) at a time" if batch > 1 else f"{conc} stream(s), one pair each",
           "ms": 1e3 * (time.perf_counter() - t0),
           "max_abs_pose_error": err, "identity_guess_error": ident}
    (log or emit_line)(rec)
    return rec


def self_launch(a):
    """`python run_segments.py --ranks N` with no launcher around it: become the launcher (VERDICT r4 item 1 -- a bare command must
    never turn into a one-rank walk of an N-segment job)."""
    import socket
    import subprocess
    if not a.launch_check and not a.one_device:
        ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if ndev < a.ranks:
            emit_line({"phase": "error", "error": f"--ranks {a.ranks} but {ndev} GPU(s) visible to this process: refusing to start"})
            raise SystemExit(3)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def launch_check(a):
    """The process group and the merge tree's edges, nothing else: rank ids all_gathered, `segments.link_selftest` along every
    (dst, src) pair of every level, verdicts combined by the MIN all-reduce the real run uses.  Host tensors under gloo."""
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if a.ranks and a.ranks != world:
        if rank == 0:
            emit_line({"phase": "error", "error": f"--ranks {a.ranks} but WORLD_SIZE={world}"})
        raise SystemExit(2)
    on_dev = a.backend == "nccl"
    dev = torch.device("cuda", 0 if a.one_device else int(os.environ.get("LOCAL_RANK", "0"))) if on_dev else torch.device("cpu")
    if on_dev:
        torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(a.backend, device_id=dev if on_dev else None)
    ids = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(ids, torch.tensor([rank], dtype=torch.int64, device=dev))
    tr = segments.DistTransport()
    st = segments.link_selftest(tr, segments.merge_schedule(world), dev)
    flag = torch.tensor([1.0 if st["ok"] else 0.0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.barrier()
    if rank == 0:
        emit_line({"phase": "launch_check", "world": world, "backend": a.backend, "ranks_seen": sorted(int(t.item()) for t in ids),
                   "selftest_ok": bool(flag.item() >= 1.0), "edges_of_rank0": st["pairs"]})
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--local", action="store_true", help="walk the whole tree on one GPU (no process group)")
    ap.add_argument("--segments", type=int, default=8, help="leaf segments for --local (otherwise WORLD_SIZE)")
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--width", type=int, default=980)
    ap.add_argument("--height", type=int, default=545)
    ap.add_argument("--gt-gaussians", type=int, default=400_000)
    ap.add_argument("--leaf-gaussians", type=int, default=200_000)
    ap.add_argument("--leaf-iters", type=int, default=30)
    ap.add_argument("--phase1-iters", type=int, default=5)
    ap.add_argument("--phase2-iters", type=int, default=10)
    ap.add_argument("--importance-views", type=int, default=0)
    ap.add_argument("--densify", action="store_true")
    ap.add_argument("--pose-lr", type=float, default=None)
    ap.add_argument("--fit-pose", action="store_true", help="refine every frame's pose while training on it (the reference's "
                                                            "camera_optimizer): one pose-step kernel per train step")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--stage-a", type=int, nargs=3, metavar=("GAUSSIANS", "IMAGE_ITERS", "POSE_ITERS"), default=None,
                    help="run stage A first (relative pose of every consecutive frame pair: single-image 3DGS of frame p, then the "
                         "SE(3) fit on frame p+1; pairs round-robin over the ranks, one all_gather) and chain ITS poses in stage B "
                         "instead of the synthetic ground truth.  The reference's counts are 1000 and 300 iterations")
    ap.add_argument("--sh-up-every", type=int, default=1000, help="raise the active SH degree after every this many global iterations of a "
                                                                  "model (the reference: 1000; a new leaf starts at degree 0)")
    ap.add_argument("--stage-a-batch", type=int, default=0, help="frame pairs of stage A fitted in one launch chain (GsrBatch); 0 = automatic "
                                                                "(4 per chain, two chains at a time; 8 when the host has CPUs for one launching "
                                                                "thread only), 1 = one pair per chain, two chains at a time (round 2)")
    ap.add_argument("--one-device", action="store_true", help="every rank on cuda:0 (with --backend gloo: the multi-process walk on a "
                                                              "one-GPU box; messages are staged through host memory)")
    ap.add_argument("--ranks", type=int, default=0, help="as a BARE command (WORLD_SIZE unset): start this many ranks of this very script "
                                                         "under torch.distributed.run (one per GPU, rendezvous on 127.0.0.1) and exit with their code; "
                                                         "fewer visible devices than ranks: a JSON error line and exit code 3")
    ap.add_argument("--launch-check", action="store_true", help="launcher plumbing only (needs no GPU with --backend gloo): form the process "
                                                                "group, run the link self-test along every edge of the merge tree, report, leave")
    a = ap.parse_args()
    if a.ranks > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)
    if a.launch_check:
        return launch_check(a)
    cfg = HTConfig(frames=a.frames, width=a.width, height=a.height, gt_gaussians=a.gt_gaussians, leaf_gaussians=a.leaf_gaussians,
                   leaf_iters_per_frame=a.leaf_iters, phase1_iters_per_frame=a.phase1_iters, phase2_iters_per_frame=[a.phase2_iters] * 3,
                   importance_views=a.importance_views, densify=a.densify, fit_pose=a.fit_pose, stage_a_batch=a.stage_a_batch, sh_up_every=a.sh_up_every)
    if a.pose_lr is not None:
        cfg.pose_lr = a.pose_lr
    if not torch.cuda.is_available():
        raise SystemExit("run_segments.py needs a ROCm GPU (no CPU fallback in the product path)")
    host_mod.cap_host_threads()   # the container's CPU quota, not the visible core count (host.py)
    if a.local:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        seq = sequence.FrameSequence(cfg.frames, cfg.gt_gaussians, cfg.width, cfg.height, dev, seed=cfg.seed)
        t0 = time.perf_counter()
        if a.stage_a:
            run_stage_a_on(seq, cfg, dev, a.stage_a, 0, 1)
        root, _ = run_local(a.segments, seq, cfg, dev)
        emit_line({"phase": "done", "world": a.segments, "mode": "local", "gaussians": root.seg.params.num_points,
                   "psnr": root.evaluate(), "total_s": time.perf_counter() - t0})
        return
    import torch.distributed as dist
    rank, world, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    if a.ranks and a.ranks != world:
        if rank == 0:
            emit_line({"phase": "error", "error": f"--ranks {a.ranks} but WORLD_SIZE={world}"})
        raise SystemExit(2)
    if not a.one_device and torch.cuda.device_count() <= local_rank:
        emit_line({"phase": "error", "rank": rank, "error": f"LOCAL_RANK={local_rank} has no device: {torch.cuda.device_count()} GPU(s) visible"})
        raise SystemExit(3)
    if a.one_device and a.backend == "nccl":
        raise SystemExit("--one-device needs --backend gloo (RCCL refuses two ranks on one device)")
    dev = torch.device("cuda", 0 if a.one_device else local_rank)
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(a.backend, device_id=dev if a.backend == "nccl" else None)
    seq = sequence.FrameSequence(cfg.frames, cfg.gt_gaussians, cfg.width, cfg.height, dev, seed=cfg.seed)
    rr = RankRunner(rank, world, segments.DistTransport(host_staging=(a.backend != "nccl")), seq, cfg, dev)
    t0 = time.perf_counter()
    if a.stage_a:   # (gloo gathers host tensors; the table is tiny -- 192 bytes per pair)
        run_stage_a_on(seq, cfg, dev, a.stage_a, rank, world, gather_device=dev if a.backend == "nccl" else torch.device("cpu"))
    def all_ok(mine: bool) -> bool:
        flag = torch.tensor([1.0 if mine else 0.0], device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item() >= 1.0)
    rr.run(barrier=dist.barrier, all_ok=all_ok)
    dist.barrier()
    if rank == 0:
        emit_line({"phase": "done", "world": world, "mode": a.backend, "gaussians": rr.seg.params.num_points,
                   "psnr": rr.evaluate(), "total_s": time.perf_counter() - t0})
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
