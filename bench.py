#!/usr/bin/env python3
"""bench.py -- train-step throughput of the MI355X-native rasterizer on the BASELINE.json headline workload.

  python bench.py --gpus 1 --steps 30 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one training iteration on one synthetic view (syn-1M, 980x545, SH degree 3, seed = rank):
activations -> rasterize (HIP) -> 0.8 L1 + 0.2 (1-SSIM) -> backward (HIP) -> Adam(eps 1e-15), i.e. the
reference's HTGaussianTrainer.train_step (/root/reference/trainer/ht3dgs_trainer.py:81-169) without
densification.  Inputs are resident in HBM before the timed region.  N>1 = one independent segment model
per GPU ("one segment per GPU"): weak scaling, no data-path collective; value = images of all
ranks / max-over-ranks time.

Rank 0 prints ONE JSON line.  Besides the contract's keys:
  `roofline`       the per-tile forward blend kernel (the kernel BASELINE.json's north_star names): achieved =
                   algorithmic bytes (44 R_eff + 28 P + 8 T) / average launch duration measured with HIP
                   events on the launch stream inside the timed region; `peak` = 8 TB/s (vendor), `peak_measured` = the
                   device-to-device copy rate measured on this box.
  `cpu_baseline`   oracle/ (the CPU restatement, kind "port") on the host cores: all threads and ONE thread -- reported
                   beside the GPU number, never part of it.
  `dropin`         the same workload through what the UNMODIFIED reference trainer can reach: `GaussianRasterizer(...)`
                   behind torch activations + cat, torch SSIM / L1, `torch.optim.Adam(eps=1e-15)` (gaussian_model_ht.py:
                   824-880, :289; trainer/losses.py:98-136).  `value` is the build's own fused step (see config.train_step).
  `merge`          one level of the merge tree (BASELINE config 4) timed AFTER the timed steps, never part of `value`:
                   importance of each child on its home rank, un-pruned child + mask point-to-point (RCCL for N > 1; a
                   device copy at N = 1), masks applied and appended at the destination.
  `other_workloads` the other BASELINE configs on the same code (N = 1 only; `--no-extras` skips them).
  `host_us`        host time of one forward + backward call through the PyTorch extension.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

def usable_cpus():
    """Host threads this process can actually keep busy: the visible cores, capped by the container's CPU-time quota
    (cgroup v2 cpu.max / v1 cfs quota) -- on the GPU boxes 256 cores are visible but the quota is 16 CPUs, and an OpenMP team
    of 256 under a 16-CPU quota spends its time throttled."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.999)))
    return n, quota



# The host-side thread pools (OpenMP in torch's CPU ops, which build the synthetic scenes) are sized to what the container may
# use: a team of 256 spinning under a 16-CPU quota exhausts the cgroup's CPU time and the kernel then throttles EVERY thread
# of the process -- including the one that launches kernels -- for the rest of the 100 ms period (seen as 4-70 ms stalls of
# single steps).  Respect an explicit OMP_NUM_THREADS of the caller.
_CPUS, _CPU_QUOTA = usable_cpus()
# (N ranks of one node share the quota: each builds its synthetic scene with its share of the threads -- eight teams of sixteen under a
#  16-CPU quota would throttle one another's launching threads exactly as described above)
_RANK_CPUS = max(1, _CPUS // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)))
os.environ.setdefault("OMP_NUM_THREADS", str(_RANK_CPUS))
os.environ.setdefault("MKL_NUM_THREADS", str(_RANK_CPUS))

import torch  # noqa: E402

try:
    torch.set_num_threads(min(torch.get_num_threads(), _RANK_CPUS))
except Exception:
    pass

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
STAGES = ["preprocess_fwd", "sort_depth", "scan", "emit", "sort_tile", "ranges", "blend_fwd", "blend_bwd", "preprocess_bwd", "cut_repair"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=980)
    ap.add_argument("--height", type=int, default=545)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--clustered", action="store_true", help="load-imbalance variant of the scene (synthetic.make_scene)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip dropin / other_workloads / merge / copy ceiling")
    ap.add_argument("--views", type=int, default=8, help="posed views the timed loop rotates through (1 = the same view every step)")
    ap.add_argument("--no-densify-stats", action="store_true", help="leave the reference's per-iteration densification statistics "
                    "(max_radii2D / xyz_gradient_accum / denom, ht3dgs_trainer.py:141-147) out of the step")
    ap.add_argument("--no-prepare-next", action="store_true", help="do not run the next render's preprocess inside the backward "
                    "(\"prepare in backward\": the step then launches k_preprocess at the start of every forward)")
    ap.add_argument("--fwd-ppt", type=int, default=0)
    ap.add_argument("--bwd-ppt", type=int, default=0)
    ap.add_argument("--tile-map", type=int, default=-1, help="2 = 2x2 tile blocks interleaved over the XCDs (default), 1 = single tiles "
                    "interleaved, 0 = one band of tiles per XCD")
    ap.add_argument("--sort-algo", type=int, default=-1, help="2 = onesweep for both sorts (default), 1 = onesweep depth sort only, 0 = hist+scan+scatter per pass")
    return ap.parse_args()


def read_profile(lib, names):
    out = {}
    for n in names:
        tot, cnt = C.c_double(0), C.c_int64(0)
        lib.gsr_profile_read(n.encode(), C.byref(tot), C.byref(cnt))
        out[n] = (tot.value, cnt.value)
    return out


def cpu_baseline(scene, threads):
    """oracle/ restatement (binary32 build) on the host cores: full forwards (K1-K6) of the same workload plus the
    K1-K5 (preprocess + duplicate + sort + ranges) leg on its own -- on all host threads (bounded to ~10 s) and on ONE
    thread (one repetition, ~8 s)."""
    from oracle import binding
    o = binding.OracleRender(means3D=scene["means3D"], opacities=scene["opacities"], viewmatrix=scene["viewmatrix"],
                             projmatrix=scene["projmatrix"], campos=scene["campos"], bg=scene["bg"],
                             image_height=scene["image_height"], image_width=scene["image_width"],
                             tanfovx=scene["tanfovx"], tanfovy=scene["tanfovy"], sh_degree=scene["sh_degree"],
                             shs=scene["shs"], scales=scene["scales"], rotations=scene["rotations"], precision="f32")
    try:
        C.CDLL("libgomp.so.1").omp_set_num_threads(threads)
    except Exception:
        pass
    t_pre, t_full, reps, R = 0.0, 0.0, 0, 0
    while t_pre + t_full < 10.0 and reps < 64:
        t0 = time.perf_counter()
        R = binding.run_stages(o, False)
        t1 = time.perf_counter()
        binding.run_stages(o, True)
        t2 = time.perf_counter()
        t_pre += t1 - t0
        t_full += t2 - t1
        reps += 1
    t_pre /= reps
    t_full /= reps
    res = {"value": 1.0 / t_full, "unit": "images/s", "cores": threads, "kind": "port",
           "sample": f"{reps} forward renders (K1-K6: preprocess+duplicate+sort+ranges+blend; no backward / loss / Adam) of the "
                     f"same syn workload, R={R} before exact tile culling, OpenMP over {threads} host threads, binary32 "
                     f"oracle/gsr_oracle.c; about {reps * (t_pre + t_full):.0f} s of wall time",
           "k1_k5_preprocess_sort_images_per_s": 1.0 / t_pre, "k1_k5_seconds": t_pre, "k1_k6_seconds": t_full}
    try:        # the single-thread leg (SURVEY.md 8d "(a) single-thread"): same code, OpenMP team of one
        gomp = C.CDLL("libgomp.so.1")
        gomp.omp_set_num_threads(1)
        t0 = time.perf_counter()
        binding.run_stages(o, False)
        t1 = time.perf_counter()
        binding.run_stages(o, True)
        t2 = time.perf_counter()
        gomp.omp_set_num_threads(threads)
        res["single_thread"] = {"value": 1.0 / (t2 - t1), "unit": "images/s", "cores": 1, "k1_k5_seconds": t1 - t0,
                                "k1_k6_seconds": t2 - t1, "sample": "1 forward render of the same workload on one host thread"}
    except Exception as e:
        res["single_thread"] = {"value": None, "sample": f"failed: {e}"}
    return res


def copy_ceiling(lib, dev, nbytes=1 << 30, reps=10):
    """The practical HBM ceiling of THIS box: the library's own float4 streaming copy (gsr_stream_copy: plain, nt, nt with four
    loads in flight per lane, and persistent workgroups with eight; several grid sizes), read + write bytes / time, best form reported.  No kernel of the step can
    stream faster than this, so no `frac_of_measured` may exceed 1."""
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    a.zero_()
    st = torch.cuda.current_stream(dev).cuda_stream
    best, form = 0.0, None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for variant in (0, 1, 2, 3):
        for blocks in ((256, 512, 1024, 2048) if variant == 3 else (2048, 4096, 8192, 16384)):
            for _ in range(2):
                lib.gsr_stream_copy(a.data_ptr(), b.data_ptr(), nbytes, variant, blocks, C.c_void_p(st))
            torch.cuda.synchronize(dev)
            e0.record()
            for _ in range(reps):
                lib.gsr_stream_copy(a.data_ptr(), b.data_ptr(), nbytes, variant, blocks, C.c_void_p(st))
            e1.record()
            torch.cuda.synchronize(dev)
            gbs = 2.0 * nbytes / (e0.elapsed_time(e1) / reps * 1e-3) / 1e9
            if gbs > best:
                best, form = gbs, f"gsr_stream_copy variant {variant}, {blocks} x 256 threads, {nbytes >> 20} MiB"
    return best, form


def timed_steps(step, steps, dev):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps


def host_leg(syn, ts, dev, calls=300):
    """Cost of one forward + backward call through each binding on a scene so small (2 000 Gaussians @128x96) that the
    ~20 kernels of a call are at their launch-latency floor: wall time per call through the PyTorch extension
    (torch.ops.gsr.rasterize: C++ autograd function, at::empty allocator) and through the ctypes FFI route
    (GSR_BINDING=ctypes: Python autograd.Function, struct marshalling, Python allocator callback).  The difference is host
    work the extension removed; the extension's figure is an upper bound of its host cost (it contains the device's own
    latency floor and the one wait for the instance count)."""
    sc = syn.make_scene(2000, 128, 96, sh_degree=3, seed=0)
    p = ts.GaussianParams(sc, dev, optimizer="torch")
    st = ts.make_settings(sc, dev, 3)
    g = torch.ones(3, 96, 128, device=dev)

    def call():
        pkg = ts.render(p, st, clamp=False, fused_activations=True)
        pkg["raw_image"].backward(g)
        p.optimizer.zero_grad(set_to_none=True)

    def measure():
        for _ in range(20):
            call()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(calls):
            call()
        torch.cuda.synchronize(dev)
        return 1e6 * (time.perf_counter() - t0) / calls
    ext = measure()
    prev = os.environ.get("GSR_BINDING")
    os.environ["GSR_BINDING"] = "ctypes"
    try:
        ct = measure()
    finally:
        if prev is None:
            os.environ.pop("GSR_BINDING", None)
        else:
            os.environ["GSR_BINDING"] = prev
    return {"fwd_bwd_call_us": ext, "ctypes_fwd_bwd_call_us": ct, "calls": calls, "binding": "torch.ops.gsr.rasterize (csrc/torch_ext.cpp)",
            "note": "wall time per forward + backward call at 2 000 Gaussians @128x96 (device at its launch-latency floor): an upper "
                    "bound of the host cost per call; ctypes_* = the same through the plain-FFI route"}


def dropin_leg(ts, scene, settings, gt, dev, steps, warmup):
    """The path the unmodified reference trainer reaches: torch activations + cat -> GaussianRasterizer(...) -> clamp ->
    torch L1 + SSIM (F.conv2d) -> backward -> torch.optim.Adam(l, lr=0.0, eps=1e-15).step()."""
    p = ts.GaussianParams(scene, dev, optimizer="torch")
    f = lambda i: ts.train_step(p, settings, gt, fused_loss=False, fused_activations=False, fused_optimizer=False)
    for i in range(warmup):
        f(i)
    sec = timed_steps(f, steps, dev)
    # the rasterizer alone on this path (forward + backward of GaussianRasterizer, grads to the activated tensors)
    def fb(i):
        pkg = ts.render(p, settings, clamp=False, fused_activations=False)
        pkg["raw_image"].backward(gt)
        p.optimizer.zero_grad(set_to_none=True)
    for i in range(2):
        fb(i)
    sec_r = timed_steps(fb, steps, dev)
    del p
    return {"value": 1.0 / sec, "unit": "images/s", "ms_per_step": 1e3 * sec, "steps": steps,
            "rasterizer_fwd_bwd_ms": 1e3 * sec_r,
            "path": "GaussianRasterizer(raster_settings)(means3D, means2D, shs, opacities, scales, rotations) behind torch exp / "
                    "sigmoid / normalize / cat, torch clamp + L1 + 11x11 SSIM (F.conv2d), torch.optim.Adam(eps=1e-15, foreach) -- "
                    "no fused extension entry point is used"}


def stage_a_leg(dev, W=980, H=545):
    """One frame pair of stage A (compute_relative_pose, ht3dgs_trainer.py:336-380) on synthetic frames: the single-image model of
    ~130 k pixel-Gaussians trained on frame p, then the SE(3) pose fit on frame p + 1 -- ms per iteration of each phase, the pose
    iteration both as the one-kernel update (gsr_pose_step) and as the torch statement of the reference's loop (exponential map +
    autograd + torch.optim.Adam on six numbers)."""
    sequence = importlib.import_module("3dgs_hierarchical_training_amd.sequence")
    stage_a = importlib.import_module("3dgs_hierarchical_training_amd.stage_a")
    ts = importlib.import_module("3dgs_hierarchical_training_amd.train_step")
    seq = sequence.FrameSequence(3, 400_000, W, H, dev, seed=0)
    seq.target(0); seq.target(1)

    def timed(**kw):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        M = stage_a.fit_pair(seq, 0, dev, n_points=130_000, seed=0, **kw)
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0, M
    timed(single_image_iters=20, pose_iters=20)                                  # warm-up (kernel loads, allocator)
    timed(single_image_iters=5, pose_iters=30, fused_pose_step=False)           # ... and of torch's small kernels for the loop statement
    t_img, _ = timed(single_image_iters=300, pose_iters=0)
    t_both, M = timed(single_image_iters=300, pose_iters=200)
    t_torch, _ = timed(single_image_iters=300, pose_iters=100, fused_pose_step=False)
    T = seq.true_rel_pose(0, 1)
    return {"gaussians": 130_000, "width": W, "height": H,
            "image_iteration_ms": 1e3 * t_img / 300, "pose_iteration_ms": 1e3 * (t_both - t_img) / 200,
            "pose_iteration_ms_torch_loop": 1e3 * (t_torch - t_img) / 100,
            "pose_error_after_300_plus_200": float((M - T).abs().max()), "identity_guess_error": float((torch.eye(4) - T).abs().max()),
            "note": "per pair the reference runs up to 1000 image iterations and 300 pose iterations (ht3dgs_trainer.py:274-333); "
                    "stage A is ~70 % of a scene's render calls"}


def make_views(syn, ts, scene, dev, deg, n_views, seed=0):
    """n_views cameras on the same cloud: view 0 is the scene's own camera (the BASELINE identity-pose pinhole), the others are
    small rigid motions of it (<= 0.05 rad, ~0.05 units) -- consecutive frames of a video, as the trainer draws them -- each
    with its own U[0,1] target.  Returns [(settings, target)]."""
    W, H = int(scene["image_width"]), int(scene["image_height"])
    gen = torch.Generator().manual_seed(1000 + seed)
    out = [(ts.make_settings(scene, dev, deg), syn.target_image(W, H, seed=1).to(dev))]
    for k in range(1, n_views):
        cam = syn.make_camera(W, H, R=syn.random_rotation(gen, 0.05), t=0.05 * torch.randn(3, generator=gen))
        sc = dict(scene)
        sc.update(cam)
        out.append((ts.make_settings(sc, dev, deg), syn.target_image(W, H, seed=1 + k).to(dev)))
    return out


def autopatch_leg(ts, scene, settings, gt, dev, steps, warmup):
    """What the UNMODIFIED reference trainer reaches once `import gsr_autopatch` ran before it, driven
    the way `train_step` drives it (ht3dgs_trainer.py:102-166): `gs_render.render(cam)` -- patched: the model's RAW tensors go to
    the kernels, activations / SH concat in-kernel --, `Loss.forward` -- patched: fused L1 + SSIM --, `loss.backward()` -- the
    per-Gaussian backward kernel also computes the Adam update into the optimizer's SHADOW buffers, the model untouched --,
    `optimizer.step()` -- `torch.optim.Adam(l, lr=0.0, eps=1e-15)` came back as FusedAdam: adopts the shadows by swapping storages;
    anything the trainer does between backward() and step() keeps the reference's meaning (optim.FusedAdam, deferred application) --,
    `zero_grad`.  No reference file is edited; the hand-over of the next preprocess is not used (the trainer draws its frames at
    random).  `separate_step_ms_per_step` = the same with GSR_AUTOPATCH_DEFERRED=0 (gradients to .grad, one-launch FusedAdam.step()).
    `with_bookkeeping` adds what the trainer runs under no_grad between backward and step on a densifying iteration: psnr
    (patched: gsr_psnr), the max_radii2D update (the trainer's own boolean-mask statement, on the patched render's LazyMask: one
    launch) and add_densification_stats (patched: one launch); `with_stock_bookkeeping` = the same three as the reference wrote
    them (torch psnr, a plain bool mask: three `nonzero` synchronisations each)."""
    import gsr_autopatch
    refstub = importlib.import_module("3dgs_hierarchical_training_amd.refstub")
    gsr_autopatch.apply()
    try:
        p = ts.GaussianParams(scene, dev, optimizer="torch")     # builds torch.optim.Adam(groups, lr=0.0, eps=1e-15) -- patched
        opt_cls = type(p.optimizer).__name__
        r = refstub.StubRender(p, bg=tuple(float(x) for x in settings.bg.cpu()))
        cam = refstub.StubCamera(settings.image_width, settings.image_height, settings.tanfovx, settings.tanfovy, settings.viewmatrix,
                                 settings.projmatrix, settings.campos, original_image=gt)

        class _Cfg:
            lambda_dssim, lambda_depth = 0.2, 0.0

        class _Loss:                                             # the attributes trainer.losses.Loss.forward reads
            cfg = _Cfg()
        loss_obj = _Loss()

        def stock_psnr(a, b):                                    # utils/image_utils.py:16-18
            mse = ((a - b) ** 2).view(a.shape[0], -1).mean(1, keepdim=True)
            return 20 * torch.log10(1.0 / torch.sqrt(mse))

        def step(book, stock=False):
            pkg = gsr_autopatch.render_fused(r, cam)
            d = gsr_autopatch.loss_forward(loss_obj, pkg["image"], gt)
            d["loss"].backward()
            with torch.no_grad():
                if book:
                    g = r.gaussians
                    (stock_psnr if stock else gsr_autopatch.psnr_fused)(pkg["image"], gt).mean().double()
                    vis, radii = pkg["visibility_filter"], pkg["radii"]
                    if stock:
                        vis = vis.as_subclass(torch.Tensor)
                    g.max_radii2D[vis] = torch.max(g.max_radii2D[vis], radii[vis])     # the trainer's own statement
                    if stock:
                        refstub.StubGaussians.add_densification_stats(g, pkg["viewspace_points"], vis)
                    else:
                        gsr_autopatch.add_densification_stats_fused(g, pkg["viewspace_points"], vis)
                p.optimizer.step()
                p.optimizer.zero_grad(set_to_none=True)
        for i in range(warmup):
            step(False)
        sec = timed_steps(lambda i: step(False), steps, dev)
        for i in range(2):
            step(True)
        sec_b = timed_steps(lambda i: step(True), steps, dev)
        for i in range(2):
            step(True, stock=True)
        sec_bs = timed_steps(lambda i: step(True, stock=True), steps, dev)
        prev_def = os.environ.get("GSR_AUTOPATCH_DEFERRED")
        os.environ["GSR_AUTOPATCH_DEFERRED"] = "0"
        try:
            for i in range(2):
                step(False)
            sec_sep = timed_steps(lambda i: step(False), steps, dev)
        finally:
            if prev_def is None:
                os.environ.pop("GSR_AUTOPATCH_DEFERRED", None)
            else:
                os.environ["GSR_AUTOPATCH_DEFERRED"] = prev_def
        # the legacy form of this leg (rounds 2-3): the wrapper's torch activations + cat + GaussianRasterizer, patched loss + optimizer
        def old(i):
            pkg = ts.render(p, settings, clamp=True, fused_activations=False)
            gsr_autopatch.loss_forward(loss_obj, pkg["image"], gt)["loss"].backward()
            p.optimizer.step()
            p.optimizer.zero_grad(set_to_none=True)
        for i in range(2):
            old(i)
        sec_old = timed_steps(old, steps, dev)
        # (the default route once more, behind the others: the first timed loop of a leg runs into clock ramps and a cold allocator at
        #  sub-millisecond sizes -- the same code measured 0.49 ms first and 0.42 ms last at 130 k Gaussians)
        for i in range(2):
            step(False)
        sec = min(sec, timed_steps(lambda i: step(False), steps, dev))
    finally:
        gsr_autopatch.remove()
    del p
    return {"value": 1.0 / sec, "unit": "images/s", "ms_per_step": 1e3 * sec, "steps": steps, "optimizer_class": opt_cls,
            "with_bookkeeping_ms_per_step": 1e3 * sec_b, "with_stock_bookkeeping_ms_per_step": 1e3 * sec_bs,
            "separate_step_ms_per_step": 1e3 * sec_sep,
            "render_unpatched_ms_per_step": 1e3 * sec_old,
            "path": "`import gsr_autopatch` + the unmodified trainer's calls: CF3DGS_Render.render (patched: raw parameters -> "
                    "rasterize_gaussians_raw, in-kernel exp / sigmoid / normalize / cat) -> Loss.forward (patched: fused clamp + L1 + SSIM) "
                    "-> backward (the Adam update computed into shadow buffers) -> torch.optim.Adam(...).step() (patched: FusedAdam adopts "
                    "the shadows by a storage swap); separate_step_* = GSR_AUTOPATCH_DEFERRED=0 (gradients to .grad + one-launch step); "
                    "render_unpatched_* = the same with GSR_AUTOPATCH_RENDER=0 (round 3's form of this leg)"}


def posed_frames_leg(ts, lib, scene, settings, dev, steps, warmup, frames=8):
    """The reference's calling convention, which the other legs do not have: EVERY frame is rendered through an
    identity camera, the frame's pose acts on the points (`rotate_seq`: get_xyz = P[seq_idx].retr().act(_xyz), /root/reference/scene/
    gaussian_model_ht.py:135-148 -- `points_transform` on the patched route) and the frame's pose optimizer steps after every render
    (/root/reference/trainer/ht3dgs_trainer.py:162-166); the frame is drawn at random every iteration (:497-536).  `import gsr_autopatch`
    + the trainer's calls, `frames` frames with their own targets, camera uid = frame index.  Reported with the forward blend's balanced
    placement on and off (same process, same model): ms per step, the blend kernel's own duration (its dispatch's timestamps, every third
    launch), and the view-cost cache's hit rate -- by the camera's uid (what gsr_autopatch passes) and by pose alone (GSR_AUTOPATCH_VIEW_ID=0)."""
    import math
    import random
    import gsr_autopatch
    refstub = importlib.import_module("3dgs_hierarchical_training_amd.refstub")
    pose_mod = importlib.import_module("3dgs_hierarchical_training_amd.pose")
    syn = importlib.import_module("3dgs_hierarchical_training_amd.synthetic")
    W, H = int(settings.image_width), int(settings.image_height)
    gsr_autopatch.apply()
    try:
        cams = [refstub.StubCamera(W, H, settings.tanfovx, settings.tanfovy, settings.viewmatrix, settings.projmatrix, settings.campos,
                                   uid=f, original_image=syn.target_image(W, H, seed=40 + f).to(dev)) for f in range(frames)]

        class _Cfg:
            lambda_dssim, lambda_depth = 0.2, 0.0

        class _Loss:
            cfg = _Cfg()
        loss_obj = _Loss()

        def stats():
            out = (C.c_int64 * 4)()
            lib.gsr_debug_view_cache_stats(W, H, out)
            return out[0], out[1], out[2]

        def measure(balance, by_uid, pose_fused=True):
            # every mode trains ITS OWN copy of the model from the same start, with the same frame draws: the blend's time follows the
            # model as it trains (on noise targets: the lists deepen), so only equal trajectories compare
            prev_pf = os.environ.get("GSR_AUTOPATCH_POSE_FUSED")
            os.environ["GSR_AUTOPATCH_POSE_FUSED"] = "1" if pose_fused else "0"
            p = ts.GaussianParams(scene, dev, optimizer="torch")
            r = refstub.StubRender(p, bg=tuple(float(x) for x in settings.bg.cpu()))
            g = r.gaussians
            gen = torch.Generator().manual_seed(4321)
            # the model's own statements: init_RT_seq (gaussian_model_ht.py:362-377: one LieGroupParameter(SE3(pose7)) per frame) and
            # training_setup(fit_pose=True) (:296-311: one Adam per frame over it) -- lietorch's public API as refstub states it in torch
            g.P = []
            for f in range(frames):
                w = torch.cat([0.02 * torch.randn(3, generator=gen), 0.03 * torch.randn(3, generator=gen)]) if f else torch.zeros(6)
                Mf = pose_mod.se3_exp(torch.cat([w[3:], w[:3]]).double())
                tr = float(Mf[0, 0] + Mf[1, 1] + Mf[2, 2])
                qw = math.sqrt(max(1e-12, 1.0 + tr)) / 2
                q = torch.tensor([float(Mf[2, 1] - Mf[1, 2]) / (4 * qw), float(Mf[0, 2] - Mf[2, 0]) / (4 * qw), float(Mf[1, 0] - Mf[0, 1]) / (4 * qw), qw])
                g.P.append(refstub.LieGroupParameter(refstub.SE3(torch.cat([Mf[:3, 3].float(), q.float()])[None].to(dev))))
            g.rotate_seq = True
            g.camera_optimizer = [torch.optim.Adam([{'params': [g.P[f]], 'lr': 1e-4, "name": "R"}], lr=0.0, eps=1e-15) for f in range(frames)]
            kinds = {type(o).__name__ for o in g.camera_optimizer}
            rng = random.Random(7)

            def step(i):
                f = rng.randrange(frames)
                g.seq_idx = f
                pkg = gsr_autopatch.render_fused(r, cams[f])
                gsr_autopatch.loss_forward(loss_obj, pkg["image"], cams[f].original_image)["loss"].backward()
                with torch.no_grad():
                    p.optimizer.step()
                    p.optimizer.zero_grad(set_to_none=True)
                    g.camera_optimizer[f].step()                      # ht3dgs_trainer.py:162-166: the pose's bits change after every render
                    g.camera_optimizer[f].zero_grad(set_to_none=True)
            lib.gsr_set_option(b"blend_balance", balance)
            prev = os.environ.get("GSR_AUTOPATCH_VIEW_ID")
            os.environ["GSR_AUTOPATCH_VIEW_ID"] = "1" if by_uid else "0"
            try:
                for i in range(max(warmup, 2 * frames)):
                    step(i)
                lib.gsr_set_option(b"profile", 3)
                read_profile(lib, ["blend_fwd"])
                s0 = stats()
                sec = timed_steps(step, steps, dev)
                s1 = stats()
                lib.gsr_set_option(b"profile", 0)
                tot, cnt = read_profile(lib, ["blend_fwd"])["blend_fwd"]
            finally:
                lib.gsr_set_option(b"profile", 0)
                for k, v in (("GSR_AUTOPATCH_VIEW_ID", prev), ("GSR_AUTOPATCH_POSE_FUSED", prev_pf)):
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
            del p, r, g
            look = s1[0] - s0[0]
            return {"ms_per_step": 1e3 * sec, "blend_fwd_us": (1e3 * tot / cnt) if cnt else None, "blend_launches_timed": cnt,
                    "view_cache_hit_rate": ((s1[1] - s0[1]) / look) if look else None, "view_cache_entries_in_use": s1[2],
                    "pose_optimizer": sorted(kinds)}
        out = {"frames": frames, "steps": steps,
               "balance_off": measure(0, True), "balance_on_by_uid": measure(1, True), "balance_on_by_pose": measure(1, False),
               "lietorch_chain_and_stock_adam": measure(1, True, pose_fused=False)}
        lib.gsr_set_option(b"blend_balance", 1)
    finally:
        lib.gsr_set_option(b"blend_balance", 1)
        gsr_autopatch.remove()
    out["note"] = ("identity camera for every frame, the frame's lietorch-shaped LieGroupParameter through get_xyz, camera_optimizer[f].step() after "
                   "every render, frames drawn at random.  Round 6: gsr_autopatch puts ONE autograd node (gsr::pose_matrix) where P[f].retr() "
                   "stands and hands out FusedPoseAdam for torch.optim.Adam over such a parameter: three one-wave kernels per iteration; "
                   "'lietorch_chain_and_stock_adam' = GSR_AUTOPATCH_POSE_FUSED=0, the group's chain (exponential map, product, matrix()) and "
                   "its autograd stated in torch + the stock Adam, what rounds 4-5 reported here.  Every mode trains its own copy of the model "
                   "from the same start with the same frame draws (the blend's time follows the model as it trains)")
    return out


def rccl_probe(dist, dev, world, rank, backend, payload=64 << 20):
    """Self-diagnosis of the process group for the first multi-GPU run: which ranks answered (an all_gather of rank ids), the
    backend, and the point-to-point rate of every level-0 merge pair (2k <-> 2k+1, 64 MiB each way, all pairs at once -- on the
    xGMI mesh every pair has its own link)."""
    if dist is None:
        return {"world": 1, "ranks_seen": [0], "backend": None, "link_GBps": {}, "note": "single process: no process group"}
    ids = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(ids, torch.tensor([rank], dtype=torch.int64, device=dev))
    seen = sorted(int(t.item()) for t in ids)
    if world == 1:
        # GSR_BENCH_FORCE_DIST=1 on one GPU: the process group is real (backend nccl = RCCL), so the collectives this code uses
        # anywhere (all_gather of stage A's pose rows, the MIN all-reduce of the link self-test, barrier, broadcast) execute once on
        # device tensors; point-to-point needs a second rank and stays unmeasured
        x = torch.arange(1 << 20, dtype=torch.float32, device=dev)
        y = x.clone()
        dist.all_reduce(y, op=dist.ReduceOp.SUM)
        mn = torch.tensor([7.0], device=dev)
        dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        rows = [torch.empty(5, 3, 4, 4, device=dev)]
        src = torch.randn(5, 3, 4, 4, device=dev)
        dist.all_gather(rows, src)
        bc = torch.full((1024,), 3.0, device=dev)
        dist.broadcast(bc, 0)
        dist.barrier()
        torch.cuda.synchronize(dev)
        ok = bool(torch.equal(x, y)) and float(mn) == 7.0 and bool(torch.equal(rows[0], src)) and float(bc.sum()) == 3072.0
        # point-to-point on DEVICE tensors as a self pair: one grouped isend + irecv (what segments.DistTransport issues for a message a
        # rank addresses to itself) -- the RCCL send / recv kernels execute; the "link" is this GPU's own HBM
        p2p = {"ok": False}
        try:
            a = torch.arange(payload // 4, dtype=torch.float32, device=dev)
            b = torch.zeros_like(a)
            for rep in range(3):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.irecv, b, 0)]):
                    w.wait()
                torch.cuda.synchronize(dev)
                dt = time.perf_counter() - t0
            p2p = {"ok": bool(torch.equal(a, b)), "bytes": payload, "ms": 1e3 * dt, "GBps": payload / dt / 1e9}
        except Exception as e:
            p2p = {"ok": False, "error": repr(e)[:300]}
        return {"world": 1, "ranks_seen": seen, "all_ranks_present": seen == [0], "backend": dist.get_backend(), "collectives_ok": ok,
                "collectives": ["all_gather", "all_reduce(SUM)", "all_reduce(MIN)", "broadcast", "barrier"], "link_GBps": {},
                "p2p_self_pair": p2p,
                "note": "process group forced at world 1 (GSR_BENCH_FORCE_DIST=1): collectives on device tensors returned, and the "
                        "point-to-point kernels ran as a self pair (batch_isend_irecv of one isend + one irecv to rank 0); a second "
                        "GPU and N > 1 stay unmeasured"}
    host = backend != "nccl"
    peer = rank ^ 1
    n = payload
    mine = torch.full((n,), rank & 0xff, dtype=torch.uint8, device="cpu" if host else dev)
    theirs = torch.empty_like(mine)
    rate, ok = 0.0, 1.0
    if peer < world:
        for rep in range(3):                    # first pass sets the channel up
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            if rank < peer:
                dist.send(mine, peer); dist.recv(theirs, peer)
            else:
                dist.recv(theirs, peer); dist.send(mine, peer)
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            rate = 2.0 * n / (time.perf_counter() - t0) / 1e9
        ok = 1.0 if bool((theirs == (peer & 0xff)).all().item()) else 0.0
    v = torch.tensor([rate, ok], dtype=torch.float64, device=dev)
    allv = [torch.zeros_like(v) for _ in range(world)]
    dist.all_gather(allv, v)
    links = {f"{r}<->{r ^ 1}": float(allv[r][0]) for r in range(0, world, 2) if (r ^ 1) < world}
    return {"world": world, "ranks_seen": seen, "all_ranks_present": seen == list(range(world)), "backend": backend,
            "payload_ok": all(float(x[1]) == 1.0 for x in allv), "link_GBps": links,
            "note": "64 MiB send + recv per level-0 merge pair, all pairs concurrently; rate = both directions / wall time of the pair"}


def stage_a_batched_leg(dev, W=980, H=545, B=8, iters=150):
    """Stage A's image iteration with B frame pairs per launch chain (batched.BatchedGaussianParams / GsrBatch) against one pair per
    chain: ms per pair and iteration, SH degree 0 with 16 coefficients stored, ~130 k pixel-Gaussians per model."""
    sequence = importlib.import_module("3dgs_hierarchical_training_amd.sequence")
    ts = importlib.import_module("3dgs_hierarchical_training_amd.train_step")
    bt = importlib.import_module("3dgs_hierarchical_training_amd.batched")
    seq = sequence.FrameSequence(B + 1, 400_000, W, H, dev, seed=0)
    ident1 = ts.with_sh_degree(seq.settings_for_pose(torch.eye(4)), 0)
    scenes = [seq.pixel_scene(p, stride=2, seed=0) for p in range(B)]
    out = {"gaussians_per_model": int(scenes[0]["means3D"].shape[0]), "width": W, "height": H, "sh_degree": 0, "models_per_chain": B}
    for name, nb in (("one_pair_per_chain", 1), ("batched", B)):
        if nb == 1:
            p = ts.GaussianParams(scenes[0], dev)
            st, tgt = ident1, seq.target(0)
        else:
            p = bt.BatchedGaussianParams(scenes, dev)
            st, tgt = bt.batch_settings([ident1] * B, dev), torch.stack([seq.target(k) for k in range(B)])
        p.active_sh_degree = 0
        f = lambda i: ts.train_step(p, st, tgt, next_settings=st)
        for i in range(20):
            f(i)
        out[name + "_ms_per_pair_iteration"] = 1e3 * timed_steps(f, iters, dev) / nb
        del p
    out["note"] = ("the reference fits the F - 1 frame pairs of stage A one after the other (ht3dgs_trainer.py:697-698), each a chain of ~17 dependent "
                   "kernels; batched = B models in one parameter store, one chain per step, every model bit-identical with training it alone; "
                   "39 pairs x (1000 + 300) iterations on one GPU: profiles/r03_stage_a_batched.txt")
    return out


_densify_warmed = False


def warm_densify(syn, ts, dm, dev):
    """One untimed clone + split + prune on a throw-away model: the first use of torch's masked-index / randn / bmm kernels loads
    their code objects and initialises rocBLAS -- 0.2-0.6 s on a box whose page cache is cold, which a 300-step leg would report as
    +2 ms per step (measured: the three densifications of the C3 leg cost 570 + 14 + 266 ms on a fresh box, 5-15 ms each warm)."""
    global _densify_warmed
    if _densify_warmed:
        return
    _densify_warmed = True
    sc = syn.make_scene(20_000, 256, 256, sh_degree=3, seed=11)
    p = ts.GaussianParams(sc, dev)
    med = float(p.get_scaling.detach().max(dim=1).values.median())
    den = dm.Densifier(p, scene_extent=med / 0.01, cfg=dm.DensifyConfig(densify_from_iter=0, densification_interval=1, percent_dense=0.01,
                                                                         opacity_reset_interval=10 ** 9, max_points=10 ** 6))
    with torch.no_grad():
        for rep in range(2):
            den.xyz_gradient_accum.fill_(1.0)
            den.denom.fill_(1.0)
            den.densify_and_prune(1e-6, 0.2, 20.0)      # half of the model is "small" (cloned), half "large" (split); low opacities pruned
    del p, den
    torch.cuda.synchronize(dev)


def workload_leg(syn, ts, raster, dev, N, W, H, deg, steps, warmup, clustered=False, densify_every=0, seed=0):
    dm = importlib.import_module("3dgs_hierarchical_training_amd.densify")
    if densify_every:
        warm_densify(syn, ts, dm, dev)
    scene = syn.make_scene(N, W, H, sh_degree=deg, seed=seed, clustered=clustered)
    gt = syn.target_image(W, H, seed=1).to(dev)
    p = ts.GaussianParams(scene, dev)
    st = ts.make_settings(scene, dev, deg)
    den = None
    if densify_every:
        den = dm.Densifier(p, scene_extent=5.0, cfg=dm.DensifyConfig(densify_from_iter=0, densification_interval=densify_every,
                                                                    densify_grad_threshold=2e-4, opacity_reset_interval=10 ** 9,
                                                                    max_points=2 * N))
    it = [0]

    def f(i):
        it[0] += 1
        ts.train_step(p, st, gt, densifier=den, iteration=it[0], next_settings=st)
    for i in range(warmup):
        f(i)
    timing = f"{steps} consecutive steps"
    if densify_every or steps < 20:
        sec = timed_steps(f, steps, dev)
    else:   # side legs of tens of steps at sub-millisecond sizes: one slow stretch on the launching thread moved them by 30 %
        h = steps // 2
        halves = [timed_steps(f, h, dev), timed_steps(f, steps - h, dev)]
        sec = min(halves)
        timing = f"the faster of two consecutive halves of {steps} steps (both in ms_per_step_halves; the headline `value` is a plain mean)"
    with torch.no_grad():
        ts.render(p, st)
    info = raster.last_call_info()
    out = {"gaussians_start": N, "gaussians_end": p.num_points, "width": W, "height": H, "sh_degree": deg, "steps": steps,
           "images_per_s": 1.0 / sec, "ms_per_step": 1e3 * sec, "timing": timing, "num_rendered_R": info["num_rendered"], "R_eff": info["staged"]}
    if not (densify_every or steps < 20):
        out["ms_per_step_halves"] = [1e3 * x for x in halves]
        out["ms_per_step_mean"] = 1e3 * (halves[0] * h + halves[1] * (steps - h)) / steps
    if densify_every:
        out["densification"] = f"every {densify_every} steps, grad threshold 2e-4 (clone + split + prune inside the timed region)"
    del p, den
    torch.cuda.empty_cache()
    return out


def knn_leg(syn, dev, sizes=(130_000, 500_000), calls=20):
    """`simple_knn._C.distCUDA2` (/root/reference/scene/gaussian_model_ht.py:20, called at every `init_model`, :211-216:
    once per leaf and twice per frame pair of stage A) on point clouds of stage A's and a leaf's size: ms per call of the HIP kernels
    (Morton sort + AABB-pruned exact 3-NN), and the reference's own fallback -- SciPy's KDTree (:31-36) -- on the host beside it."""
    import simple_knn._C as knn
    out = {}
    for n in sizes:
        pts = syn.make_scene(n, 980, 545, sh_degree=0, seed=21)["means3D"].to(dev).contiguous()
        for _ in range(3):
            knn.distCUDA2(pts)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(calls):
            d = knn.distCUDA2(pts)
        torch.cuda.synchronize(dev)
        ms = 1e3 * (time.perf_counter() - t0) / calls
        rec = {"ms_per_call": ms, "points_per_s": n / (ms * 1e-3), "calls": calls}
        if n <= 200_000:
            try:
                from scipy.spatial import KDTree
                p_np = pts.cpu().numpy()
                t0 = time.perf_counter()
                dist_, _ = KDTree(p_np).query(p_np, k=4)
                rec["scipy_kdtree_host_ms"] = 1e3 * (time.perf_counter() - t0)
                ref = (dist_[:, 1:] ** 2).mean(1)
                rec["max_rel_diff_vs_scipy"] = float(abs(d.cpu().double().numpy() - ref).max() / ref.max())
            except Exception as e:
                rec["scipy_kdtree_host_ms"] = None
                rec["scipy_error"] = repr(e)
        out[f"{n} points"] = rec
        del pts
    return out


def merge_leg(dev, dist, world, rank, params, scene, ts, syn, deg):
    """One level of the merge tree, level 0 pairs (2k <- 2k+1).  N = 1: the segment merges with a copy of itself through
    the in-process transport (a device copy stands for the link)."""
    hier = importlib.import_module("3dgs_hierarchical_training_amd.hierarchy")
    seg_mod = importlib.import_module("3dgs_hierarchical_training_amd.segments")
    W, H = int(scene["image_width"]), int(scene["image_height"])
    views = [ts.make_settings(scene, dev, deg)]
    cam = syn.make_scene(8, W, H, sh_degree=deg, seed=77, posed=True)
    sc2 = dict(scene)
    for k in ("viewmatrix", "projmatrix", "campos"):
        sc2[k] = cam[k]
    views.append(ts.make_settings(sc2, dev, deg))
    raw = params.raw()
    T = torch.eye(4)
    hier.calc_importance(raw, views[:1])      # warm the non-fused backward variant
    if world == 1:
        best = None
        # with a process group at world 1 (GSR_BENCH_FORCE_DIST=1, backend nccl) the child travels through RCCL's point-to-point
        # kernels as a self pair (segments.DistTransport); without one, through the in-process mailbox (a device copy)
        over_rccl = dist is not None and dist.get_backend() == "nccl"
        for rep in range(2):          # the first pass pays torch's one-time kernel loads (topk, masked gathers); report the second
            tr = seg_mod.DistTransport() if over_rccl else seg_mod.LocalTransport(2)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            if not over_rccl:
                tr.rank = 1
            s = hier.merge_send(tr, 0, raw, views, 0.5)
            if not over_rccl:
                tr.rank = 0
            d = hier.merge_recv(tr, 0 if over_rccl else 1, raw, views, 0.5, T)
            torch.cuda.synchronize(dev)
            total = 1e3 * (time.perf_counter() - t0)
            best = {"merge_ms": total, "merge_bytes": s["bytes"], "pairs": 1,
                    "transport": "RCCL send / recv kernels, self pair on one GPU (N = 1)" if over_rccl else "in-process device copy (N = 1)",
                    "importance_views": len(views), "importance_ms_src": s["importance_ms"], "importance_ms_dst": d["importance_ms"],
                    "send_ms": s["send_ms"], "recv_ms": d["recv_ms"], "append_ms": d["append_ms"], "gaussians_child": s["n"],
                    "gaussians_merged": d["n_merged"], "note": "both children's importance run back to back on the one GPU"}
            del d
        return best
    tr = seg_mod.DistTransport(host_staging=(dist.get_backend() != "nccl"))
    pairs = seg_mod.merge_schedule(world)[0]
    role = seg_mod.partner(rank, pairs)
    for rep in range(2):              # first pass: torch's one-time kernel loads + RCCL's channel set-up; the second is reported
        torch.cuda.synchronize(dev)
        dist.barrier()
        t0 = time.perf_counter()
        st = {"importance_ms": 0.0, "xfer_ms": 0.0, "bytes": 0.0, "append_ms": 0.0, "n_merged": 0.0}
        # phase 1, local: every paired rank scores its own child.  A rank that fails here must not leave its partner
        # blocked in a point-to-point call, so the ranks agree (one MIN all-reduce of a flag) before any send / recv
        drop, ok = None, 1.0
        try:
            if role is not None:
                drop = hier.prune_mask(hier.calc_importance(raw, views), 0.5)
            torch.cuda.synchronize(dev)
        except Exception:
            ok = 0.0
        st["importance_ms"] = 1e3 * (time.perf_counter() - t0)
        flag = torch.tensor([ok], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if float(flag.item()) < 1.0:
            raise RuntimeError("importance failed on at least one rank; merge exchange skipped on all")
        # phase 2: the exchange
        if role is not None and role[0] == "send":
            s = hier.merge_send(tr, role[1], raw, views, 0.5, drop=drop)
            st.update(xfer_ms=s["send_ms"], bytes=float(s["bytes"]))
        elif role is not None:
            d = hier.merge_recv(tr, role[1], raw, views, 0.5, T, drop=drop)
            st.update(xfer_ms=d["recv_ms"], append_ms=d["append_ms"], n_merged=float(d["n_merged"]))
            del d
        torch.cuda.synchronize(dev)
        local = 1e3 * (time.perf_counter() - t0)
    v = torch.tensor([local, st["importance_ms"], st["xfer_ms"], st["append_ms"], st["n_merged"]], device=dev, dtype=torch.float64)
    dist.all_reduce(v, op=dist.ReduceOp.MAX)
    b = torch.tensor([st["bytes"]], device=dev, dtype=torch.float64)
    dist.all_reduce(b, op=dist.ReduceOp.SUM)
    return {"merge_ms": float(v[0]), "merge_bytes": float(b[0]), "pairs": len(pairs), "transport": "RCCL send/recv, one xGMI link per pair" if dist.get_backend() == "nccl" else f"{dist.get_backend()} (test transport, host-staged)",
            "importance_views": len(views), "importance_ms_max": float(v[1]), "xfer_ms_max": float(v[2]), "append_ms_max": float(v[3]),
            "gaussians_merged_max": int(v[4]), "link_GBps": (float(b[0]) / len(pairs)) / (float(v[2]) * 1e-3) / 1e9 if float(v[2]) > 0 else None}


METRIC = "train-step images/sec + fwd+bwd ms @1M Gaussians, 980x545; 1/2/4/8 GPU"   # BASELINE.json's metric


def error_line(args, world, msg, code):
    """A run that cannot measure what was asked prints ONE JSON line saying so and exits non-zero: never a number of a smaller job."""
    print(json.dumps({"metric": METRIC, "value": None, "unit": "images/s", "n_gpus": args.gpus, "world_size_seen": world,
                      "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "error": msg}), flush=True)
    raise SystemExit(code)


def self_launch(args):
    """`python bench.py --gpus N` as a BARE command (no torch.distributed.run around it, WORLD_SIZE unset): this process becomes the
    launcher -- it starts `torch.distributed.run --nnodes=1 --nproc-per-node N` on this very file with the same arguments (one rank per
    GPU over RCCL, rendezvous on 127.0.0.1 and a free port), passes the ranks' output through (rank 0 prints the one JSON line) and
    exits with their code.  Fewer than N visible devices: a JSON error line and exit code 3 -- never a silent world-1 run
    (GSR_BENCH_ONE_DEVICE=1, the one-GPU test plumbing, puts every rank on cuda:0 and lifts that check)."""
    import socket
    import subprocess
    one_dev = os.environ.get("GSR_BENCH_ONE_DEVICE") == "1"
    if not LAUNCH_CHECK and not one_dev:
        ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if ndev < args.gpus:
            error_line(args, 1, f"--gpus {args.gpus} but {ndev} GPU(s) visible to this process: refusing to run a smaller job under that name", 3)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env["GSR_BENCH_SELF_LAUNCHED"] = "1"
    raise SystemExit(subprocess.call(cmd, env=env))


# GSR_BENCH_LAUNCH_CHECK=1: launcher plumbing only (tests/test_launch_cpu.py, no GPU): the ranks come up, form the process group,
# run the group's self-diagnosis and rank 0 prints a line with "value": null and "launch_check": true.  Nothing is rendered or timed.
LAUNCH_CHECK = os.environ.get("GSR_BENCH_LAUNCH_CHECK") == "1"


def launch_check(args, world, rank):
    import torch.distributed as dist
    backend = os.environ.get("GSR_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        error_line(args, world, "GSR_BENCH_LAUNCH_CHECK needs GSR_BENCH_BACKEND=gloo (it runs without a GPU)", 2)
    dist.init_process_group(backend)
    r = rccl_probe(dist, torch.device("cpu"), world, rank, backend, payload=1 << 20)
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": None, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "launch_check": True, "self_launched": os.environ.get("GSR_BENCH_SELF_LAUNCHED") == "1", "rccl": r}), flush=True)
    dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:      # a launcher that started another number of ranks than the line would claim
        if rank == 0:
            error_line(args, world, f"--gpus {args.gpus} but WORLD_SIZE={world}", 2)
        raise SystemExit(2)
    if LAUNCH_CHECK:
        return launch_check(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (no CPU fallback in the product path)")
    if os.environ.get("GSR_BENCH_ONE_DEVICE") != "1" and torch.cuda.device_count() <= local_rank:
        if rank == 0:
            error_line(args, world, f"rank with LOCAL_RANK={local_rank} has no device: {torch.cuda.device_count()} GPU(s) visible", 3)
        raise SystemExit(3)
    # GSR_BENCH_BACKEND=gloo + GSR_BENCH_ONE_DEVICE=1: every rank on cuda:0, gloo instead of RCCL (point-to-point messages staged
    # through host memory) -- lets the N > 1 code path of this file run on a one-GPU box (tests/test_gpu_segments.py); the
    # numbers of such a run mean nothing
    backend = os.environ.get("GSR_BENCH_BACKEND", "nccl")
    if os.environ.get("GSR_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("GSR_BENCH_FORCE_DIST") == "1":   # (the env var exercises the RCCL path on one GPU)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend, device_id=dev if backend == "nccl" else None)

    syn = importlib.import_module("3dgs_hierarchical_training_amd.synthetic")
    ts = importlib.import_module("3dgs_hierarchical_training_amd.train_step")
    L = importlib.import_module("3dgs_hierarchical_training_amd._lib")
    raster = importlib.import_module("3dgs_hierarchical_training_amd.rasterizer")
    lib = L.load()
    if args.fwd_ppt:
        lib.gsr_set_option(b"blend_fwd_ppt", args.fwd_ppt)
    if args.bwd_ppt:
        lib.gsr_set_option(b"blend_bwd_ppt", args.bwd_ppt)
    if args.sort_algo >= 0:
        lib.gsr_set_option(b"sort_algo", args.sort_algo)
    if args.tile_map >= 0:
        lib.gsr_set_option(b"tile_map", args.tile_map)
    for kv in os.environ.get("GSR_OPTS", "").split(","):      # A/B runs: GSR_OPTS="exp_bwd=1,bwd_split=8"
        if "=" in kv:
            k, v = kv.split("=")
            if lib.gsr_set_option(k.encode(), int(v)) != 0:
                raise SystemExit(f"gsr_set_option({k}, {v}) refused")

    N, W, H, deg = args.gaussians, args.width, args.height, args.sh_degree
    scene = syn.make_scene(N, W, H, sh_degree=deg, seed=rank, clustered=args.clustered)
    params = ts.GaussianParams(scene, dev)
    # the timed loop rotates through `--views` posed cameras of the cloud (a trainer draws a different frame every step): the
    # speculative binning's capacity hint and the hand-over to the next step see the camera change every step
    views = make_views(syn, ts, scene, dev, deg, max(1, args.views), seed=rank)
    settings, gt = views[0]
    V = len(views)
    dm = importlib.import_module("3dgs_hierarchical_training_amd.densify")
    # the reference's train_step collects the densification statistics after every backward (ht3dgs_trainer.py:137-147); here they
    # accumulate inside the per-Gaussian backward kernel.  (No clone / split / prune in the headline: that is the C3 leg.)
    den = None if args.no_densify_stats else dm.Densifier(params, scene_extent=5.0, cfg=dm.DensifyConfig(
        densify_from_iter=10 ** 9, opacity_reset_interval=10 ** 9, densify_until_iter=10 ** 9))

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    it_count = [0]

    def one_step():
        i = it_count[0]
        it_count[0] += 1
        st, tgt = views[i % V]
        # the next step's camera is known (frames are drawn ahead): its preprocess rides in this step's backward
        nxt = None if args.no_prepare_next else views[(i + 1) % V][0]
        ts.train_step(params, st, tgt, next_settings=nxt, densifier=den, iteration=i + 1)

    for _ in range(args.warmup):
        one_step()
    # instance statistics from one un-timed forward (they do not change the timed work)
    with torch.no_grad():
        pkg = ts.render(params, settings)
    n_visible = int((pkg["radii"] > 0).sum().item())
    del pkg
    # timed region: only the forward blend kernel (the roofline kernel) is timed, by the start / stop timestamps of its own dispatch, and
    # only on every third step -- a timed launch idles the queue ~11 us (tools/api_timeline.sh), all of it inside `value`; three is coprime
    # with the eight rotating views, so every view is sampled.  The per-stage breakdown comes from a few extra UNTIMED steps afterwards
    lib.gsr_set_option(b"profile", 3)
    read_profile(lib, STAGES)  # drop anything recorded so far
    sync_all()
    counters = ("forward_calls", "forward_ns", "forward_wait_ns", "backward_calls", "backward_ns", "spec_overflows", "spec_forwards", "exact_forwards")
    c0 = {k: lib.gsr_get_counter(k.encode()) for k in counters}

    def cut_stats():
        out = (C.c_int64 * 5)()
        return [int(v) for v in out] if lib.gsr_debug_list_cut_stats(W, H, out) == 0 else [0] * 5
    lc0 = cut_stats()      # (synchronises the device: outside the timed region)
    t0 = time.perf_counter()
    stamps = [t0]
    for _ in range(args.steps):
        one_step()
        stamps.append(time.perf_counter())   # host clock only (each step already waits for the forward's instance count)
    sync_all()
    elapsed = time.perf_counter() - t0
    c1 = {k: lib.gsr_get_counter(k.encode()) for k in c0}
    lc1 = cut_stats()
    lib.gsr_set_option(b"profile", 0)
    prof_blend = read_profile(lib, ["blend_fwd"])["blend_fwd"]
    # R_eff of every view (the roofline's algorithmic bytes are the mean over the views the timed launches rendered)
    r_eff_views, r_views = [], []
    with torch.no_grad():
        for st, _ in views:
            ts.render(params, st)
            info = raster.last_call_info()
            r_views.append(info["num_rendered"]); r_eff_views.append(info["staged"])
    lib.gsr_set_option(b"profile", 1)
    for _ in range(min(5, args.steps)):
        one_step()
    torch.cuda.synchronize(dev)
    lib.gsr_set_option(b"profile", 0)
    prof = read_profile(lib, STAGES)
    prof["blend_fwd"] = prof_blend   # the figure the roofline uses: measured inside the timed region
    per_rank_ms = None
    if dist is not None:
        # MAX over ranks, as the contract asks -- taken from ONE all_gather of the ranks' own times, so that the line also shows every
        # rank's ms per step and the slowest / fastest ratio (the first real 8-GPU run shows stragglers by itself)
        tdev = torch.device("cpu") if dist.get_backend() == "gloo" else dev
        t = torch.tensor([elapsed], device=tdev, dtype=torch.float64)
        allt = [torch.empty(1, device=tdev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank = [float(x.item()) for x in allt]
        per_rank_ms = [1000.0 * x / args.steps for x in per_rank]
        elapsed = max(per_rank)
    rccl = None     # filled under the watchdog below, after the throughput result is safe

    # host-side work of the library per forward + backward in the timed steps (counters of gsr_get_counter, read after them)
    lib_host = None
    if c1["forward_calls"] > c0["forward_calls"]:
        nf, nbk = c1["forward_calls"] - c0["forward_calls"], max(1, c1["backward_calls"] - c0["backward_calls"])
        lib_host = {"gsr_forward_us": 1e-3 * ((c1["forward_ns"] - c0["forward_ns"]) - (c1["forward_wait_ns"] - c0["forward_wait_ns"])) / nf,
                    "gsr_forward_wait_us": 1e-3 * (c1["forward_wait_ns"] - c0["forward_wait_ns"]) / nf,
                    "gsr_backward_us": 1e-3 * (c1["backward_ns"] - c0["backward_ns"]) / nbk}

    # R and R_eff: mean over the views the loop rotates through (state after the timed steps)
    R, R_eff = int(sum(r_views) / len(r_views)), int(sum(r_eff_views) / len(r_eff_views))

    # one merge level (config 4), outside the timed region; every rank takes part.  It is the only place where ranks exchange
    # data, so it runs under a watchdog: if the exchange has not come back after two minutes, rank 0 prints the line without
    # it and every rank leaves -- a stuck link must not cost the throughput measurement that is already complete.
    merge = None
    state = {"res": None, "printed": False}

    def bail():
        if rank == 0 and state["res"] is not None and not state["printed"]:
            if state["res"].get("rccl") is None:
                state["res"]["rccl"] = {"world": world, "error": "the process-group probe did not complete within 120 s (watchdog)"}
            state["res"]["merge"] = {"merge_ms": None, "error": "merge exchange did not complete within 120 s (watchdog)"}
            print(json.dumps(state["res"]), flush=True)
        os._exit(0)

    import threading
    dog = None
    if world > 1:        # everything behind the timed steps that talks to another rank runs under this watchdog
        dog = threading.Timer(120.0, bail)
        dog.daemon = True

    def run_probe():
        nonlocal rccl
        try:
            rccl = rccl_probe(dist, dev, world, rank, backend)
        except Exception as e:   # a diagnostic must never take the bench line down
            rccl = {"world": world, "error": repr(e)}
        if state["res"] is not None:
            state["res"]["rccl"] = rccl

    def run_merge():
        nonlocal merge
        want = os.environ.get("GSR_BENCH_MERGE")        # "1": also with --no-extras; "0": never
        if want == "0" or (args.no_extras and want != "1"):
            return
        try:
            merge = merge_leg(dev, dist, world, rank, params, scene, ts, syn, deg)
        except Exception as e:   # a diagnostic must never take the bench line down
            merge = {"merge_ms": None, "error": repr(e)}

    if rank != 0:
        if dog is not None:
            dog.start()
        run_probe()
        run_merge()
        if dog is not None:
            dog.cancel()
        if dist is not None:
            dist.destroy_process_group()
        return
    P = W * H
    T = ((W + 15) // 16) * ((H + 15) // 16)
    ms_per_step = 1000.0 * elapsed / args.steps
    # host-side enqueue intervals of the timed steps (a stall of the launching thread shows up as max >> median)
    gaps = sorted(1e3 * (b - a) for a, b in zip(stamps[:-1], stamps[1:]))
    step_host = {"median": gaps[len(gaps) // 2], "p90": gaps[min(len(gaps) - 1, (9 * len(gaps)) // 10)], "max": gaps[-1]} if gaps else {}
    stage_ms = {k: (v[0] / v[1] if v[1] else None) for k, v in prof.items()}
    fwd_stages = ["preprocess_fwd", "sort_depth", "scan", "emit", "sort_tile", "ranges", "blend_fwd"]
    bwd_stages = ["blend_bwd", "preprocess_bwd"]
    fwd_ms = sum(stage_ms[k] or 0.0 for k in fwd_stages)
    bwd_ms = sum(stage_ms[k] or 0.0 for k in bwd_stages)      # both backward stages (the in-kernel Adam rides in the second)
    blend_ms = stage_ms["blend_fwd"]
    alg_bytes = 44.0 * R_eff + 28.0 * P + 8.0 * T
    achieved = (alg_bytes / (blend_ms * 1e-3) / 1e9) if blend_ms else None
    # HBM bytes per launch and the instruction counters from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    # runs, gfx950 read-side x2 correction; tools/pmc_profile.sh + tools/pmc_summary.py).  PMC collection cannot run inside this
    # process, so the figures are taken from the newest committed summary measured on this very workload (named in the line).
    cal = {"plain": 2.366, "double_pass": 4.296, "trans": 8.152, "single_wave": 5.754}
    cal_src = None
    try:
        cj = json.load(open(os.path.join(REPO, "profiles", "r03_valu_calib.json")))
        cal, cal_src = cj["classes"], "profiles/r03_valu_calib.json (tools/microbench/valu_calib.hip on MI355X)"
    except Exception:
        pass
    pmc, pmc_src = {}, None
    if (N, W, H, deg, args.clustered) == (1_000_000, 980, 545, 3, False):
        for name in ("r06_pmc_blend.json", "r05_pmc_blend.json", "r04_pmc_blend.json", "r03_pmc_blend.json", "r02_pmc_blend.json", "r01_pmc_blend.json"):
            pmc_file = os.path.join(REPO, "profiles", name)
            if not os.path.exists(pmc_file):
                continue
            try:
                pmc, pmc_src = json.load(open(pmc_file))["kernels"], f"profiles/{name}"
                break
            except Exception:
                continue

    def valu_roof(kernel_substr, launch_ms):
        """The vector-pipe roof of a kernel: wave instructions per launch (PMC) x calibrated SIMD cycles per instruction against the
        SIMD cycles the launch had (1024 SIMDs x duration x clock).  A lower bound of the pipe's busy fraction: packed-f32 / DPP /
        f64 instructions cost `double_pass` cycles but are counted at `plain`."""
        kv = next((v for k, v in pmc.items() if kernel_substr in k), None)
        if kv is None or "SQ_INSTS_VALU" not in kv:
            return None
        trans = kv.get("SQ_INSTS_VALU_TRANS_F32", kv.get("SQ_INSTS_VALU_TRANS", 0.0))
        cyc = ((kv["SQ_INSTS_VALU"] - trans) * cal["plain"] + trans * cal["trans"]) / 1024.0
        avail = kv["GRBM_GUI_ACTIVE"] / 8.0 if "GRBM_GUI_ACTIVE" in kv else None
        return {"valu_wave_instructions_per_launch": kv["SQ_INSTS_VALU"], "of_them_transcendental": trans,
                "salu_wave_instructions_per_launch": kv.get("SQ_INSTS_SALU"),
                "cycles_per_instruction": {"plain": cal["plain"], "trans": cal["trans"], "double_pass_not_separable": cal["double_pass"],
                                           "one_wave_alone": cal["single_wave"], "source": cal_src},
                "valu_pipe_cycles_per_simd": cyc, "simd_cycles_available": avail, "frac": (cyc / avail) if avail else None,
                # the launch's time at calibrated full vector issue (PMC instruction counts of the committed
                # summary x calibrated cycles / 1024 SIMDs, at 2.4 GHz) over THIS run's measured launch time
                "valu_roof_ms": cyc / 2.4e6, "frac_of_valu_roof": (cyc / 2.4e6 / launch_ms) if launch_ms else None,
                # SQ_WAVE_CYCLES counts resident wave time in units of 4 cycles, summed over the chip
                "avg_waves_per_simd": (kv["SQ_WAVE_CYCLES"] * 4.0 / (1024.0 * avail)) if ("SQ_WAVE_CYCLES" in kv and avail) else None,
                # the hardware's own count of the cycles a SIMD's vector pipe was executing (SQ_ACTIVE_INST_VALU, units of 4 cycles,
                # summed over the 1024 SIMDs) against the launch: what the calibrated estimate above bounds from below
                "valu_busy_frac_counted": (kv["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * avail)) if ("SQ_ACTIVE_INST_VALU" in kv and avail) else None,
                "counters_from": pmc_src, "counters_launch_ms": (avail / 2.4e6) if avail else None, "this_run_launch_ms": launch_ms}

    def counted_traffic(kernel_substr, launch_ms):
        """(traffic, reason): the HBM bytes of the committed PMC summary -- withheld (None + the reason) when that collection's launch
        time and THIS run's differ by more than 10 %: counters of another tree / another box must not stand next to this run's time
        (VERDICT r5 weak #12)."""
        kv = next((v for k, v in pmc.items() if kernel_substr in k), {})
        t = kv.get("hbm_traffic_bytes")
        if t is None:
            return None, None
        cms = (kv["GRBM_GUI_ACTIVE"] / 8.0 / 2.4e6) if "GRBM_GUI_ACTIVE" in kv else None
        if cms and launch_ms and abs(cms - launch_ms) > 0.10 * launch_ms:
            return None, (f"withheld: the committed counters ({pmc_src}) were collected at {cms:.4f} ms per launch, this run measures "
                          f"{launch_ms:.4f} ms (> 10 % apart); the summary held {t:.0f} bytes")
        return t, None

    traffic, traffic_withheld = counted_traffic("k_blend_fwd", blend_ms)
    valu_fwd = valu_roof("k_blend_fwd", blend_ms)
    n_vis = n_visible
    blend_bwd_ms, preb_ms = stage_ms["blend_bwd"], stage_ms["preprocess_bwd"]
    others = []
    if blend_bwd_ms:
        ab = 44.0 * R_eff + 36.0 * P + 40.0 * n_vis
        tr_b, tr_b_why = counted_traffic("k_blend_bwd2", blend_bwd_ms)
        others.append({"kernel": "gsr::k_blend_bwd2<false>", "bound": "valu", "achieved": ab / (blend_bwd_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": ab / (blend_bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": ab,
                       "avg_launch_ms": blend_bwd_ms, "traffic": tr_b, "traffic_withheld": tr_b_why, "valu": valu_roof("k_blend_bwd2", blend_bwd_ms),
                       "note": "frac = algorithmic bytes against the HBM peak (the figure SURVEY 8d asks for); the roof that binds is the "
                               "vector pipe: see `valu` (packed f32 math + DPP butterfly: mostly double-pass instructions)"})
    if preb_ms:
        # params 236 + ggrad 48 + moments 472 in; params + moments 708 + means2D grad 12 out; + 68 out for the next render's
        # splat / radii / key / id / tile records when that render's preprocess rides along ("prepare in backward"); + 4 in (radii)
        # and 12 in / 12 out for the densification statistics of the visible Gaussians
        ab = (1476.0 + (0.0 if args.no_prepare_next else 68.0)) * N + (0.0 if den is None else 4.0 * N + 24.0 * n_vis)
        tr_p, tr_p_why = counted_traffic("k_preprocess_bwd", preb_ms)
        others.append({"kernel": "gsr::k_preprocess_bwd<3, true, false, true, 3> (per-Gaussian backward + in-kernel Adam + next preprocess)", "bound": "hbm",
                       "achieved": ab / (preb_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": ab / (preb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": ab,
                       "avg_launch_ms": preb_ms, "traffic": tr_p, "traffic_withheld": tr_p_why, "note": "durations from the untimed stage-profiling steps"})
    binds = None
    if valu_fwd and valu_fwd.get("frac") is not None:
        counted = valu_fwd.get("valu_busy_frac_counted")
        binds = (f"not HBM: the vector pipe.  By the hardware's own count (SQ_ACTIVE_INST_VALU) a SIMD's vector pipe is executing "
                 f"{counted:.2f} of the launch; instruction counts x the calibrated {cal['plain']:.2f} cycles per plain wave64 instruction "
                 f"({cal['trans']:.1f} per transcendental, {cal['double_pass']:.1f} per packed / DPP instruction, which the counters cannot separate) "
                 f"bound that from below at {valu_fwd['frac']:.2f}.  The launch averages {valu_fwd['avg_waves_per_simd']:.1f} resident waves per SIMD "
                 "(one wave per 8x8 sub-tile, all started at once: the kernel ends with its longest lists); fewer instructions per visit is "
                 "the lever that is left (DESIGN.md sections 4, 8)") if (valu_fwd.get("avg_waves_per_simd") and counted) else None
    roofline = {"kernel": "gsr::k_blend_fwd_w6<true>", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic, "traffic_withheld": traffic_withheld,
                "traffic_source": f"{pmc_src} (rocprofv3 --pmc, separate passes; a committed summary, not this run)" if traffic else None,
                "valu": valu_fwd, "binding_roof": binds,
                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": blend_ms,
                "launches_timed": prof["blend_fwd"][1], "R": R, "R_eff": R_eff, "R_eff_per_view": r_eff_views, "P": P, "T": T}
    res = {
        "metric": METRIC,
        "value": world * args.steps / elapsed, "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"syn-{N} Gaussians{' (clustered)' if args.clustered else ''}, {W}x{H}, SH degree {deg}, pinhole camera "
                               f"(FoVx {syn.FOVX_FRANCIS}); the timed loop rotates through {len(views)} posed views of the cloud (view 0 = the identity "
                               "pose of the BASELINE recipe, the others rigid motions of it by <= 0.05 rad / ~0.05 units), each with its own "
                               "U[0,1] target; one view per step per GPU",
                   "views": len(views), "densification_statistics_in_step": den is not None,
                   "gaussians": N, "width": W, "height": H, "sh_degree": deg, "visible": n_visible,
                   "num_rendered_R": R, "parallelism": f"{world} independent segment replica(s), no data-path collective",
                   "train_step": "activations + rasterize fwd + 0.8*L1+0.2*(1-SSIM) + backward + Adam(eps=1e-15) on all 59 floats "
                                 "per Gaussian (update applied inside the per-Gaussian backward kernel" +
                                 ("" if args.no_prepare_next else ", which also runs the NEXT step's preprocess on the updated parameters: "
                                  "one preprocess per step either way, k_preprocess is just not a separate launch") +
                                 "); the unmodified reference trainer reaches the `dropin` path instead"},
        "fwd_bwd_ms": fwd_ms + bwd_ms, "rasterizer_fwd_ms": fwd_ms, "rasterizer_bwd_ms": bwd_ms,
        "stage_ms": stage_ms, "step_host_ms": step_host, "roofline": roofline, "roofline_other_kernels": others,
        "spec_overflows": c1["spec_overflows"] - c0["spec_overflows"],
        # the list cut (include/gsr.h "list_cut") over the timed steps: renders that ran it, renders whose lists had to be repaired on the
        # device (a tile needed more than it kept: the counter of re-runs), tiles repaired, and per render the chunks of the depth
        # order behind the frame's cut (of `chunks`) and the deep tiles served one by one
        "list_cut": {"renders": lc1[0] - lc0[0], "renders_repaired": lc1[1] - lc0[1], "tiles_repaired": lc1[2] - lc0[2],
                     "chunks_behind_cut_per_render": (lc1[3] - lc0[3]) / max(1, lc1[0] - lc0[0]),
                     "deep_tiles_per_render": (lc1[4] - lc0[4]) / max(1, lc1[0] - lc0[0])},
        "speculative_forwards": c1["spec_forwards"] - c0["spec_forwards"], "exact_forwards": c1["exact_forwards"] - c0["exact_forwards"],
        "rccl": rccl,
    }
    if per_rank_ms is not None:
        res["per_rank_ms_per_step"] = per_rank_ms
        res["rank_time_max_over_min"] = max(per_rank_ms) / min(per_rank_ms)
    state["res"] = res
    if dog is not None:
        dog.start()
    run_probe()
    run_merge()
    if dog is not None:
        dog.cancel()
    if merge is not None:
        res["merge"] = merge
    if world == 1 and not args.no_extras:
        try:
            roofline["peak_measured"], roofline["peak_measured_form"] = copy_ceiling(lib, dev)
            roofline["frac_of_measured"] = (achieved / roofline["peak_measured"]) if achieved else None
            for o in others:
                o["peak_measured"] = roofline["peak_measured"]
                o["frac_of_measured"] = o["achieved"] / roofline["peak_measured"]
        except Exception as e:
            roofline["peak_measured"] = None
            roofline["peak_measured_error"] = repr(e)
        try:
            res["host_us"] = host_leg(syn, ts, dev)
        except Exception as e:
            res["host_us"] = {"fwd_bwd_call_us": None, "error": repr(e)}
        if lib_host is not None:      # the library's own accounting over the timed steps of the headline workload
            res["host_us"]["timed_steps"] = dict(lib_host, note="wall time inside gsr_forward (minus its wait for the instance count) and "
                                                 "gsr_backward per call in the timed steps: allocator callbacks + kernel launches; the "
                                                 "PyTorch dispatcher / autograd / loss ops around them are not in it")
        try:
            res["dropin"] = dropin_leg(ts, scene, settings, gt, dev, steps=min(args.steps, 10), warmup=2)
        except Exception as e:
            res["dropin"] = {"value": None, "error": repr(e)}
        try:
            res["dropin_autopatch"] = autopatch_leg(ts, scene, settings, gt, dev, steps=min(args.steps, 10), warmup=2)
            # ... and at what stage A of the reference runs (~70 % of a scene's render calls): one single-image model of ~130 k
            # Gaussians, active SH degree 0 with 16 coefficients stored (gaussian_model_ht.py:68), the same camera every iteration
            sc_a = syn.make_scene(130_000, W, H, sh_degree=0, seed=3)
            leg_a = autopatch_leg(ts, sc_a, ts.make_settings(sc_a, dev, 0), syn.target_image(W, H, seed=2).to(dev), dev, steps=40, warmup=10)
            res["dropin_autopatch"]["stage_a_130k_degree0"] = {k: leg_a[k] for k in ("value", "ms_per_step", "with_bookkeeping_ms_per_step",
                                                                                    "with_stock_bookkeeping_ms_per_step", "separate_step_ms_per_step", "render_unpatched_ms_per_step", "steps")}
        except Exception as e:
            res["dropin_autopatch"] = {"value": None, "error": repr(e)}
        try:
            if isinstance(res.get("dropin_autopatch"), dict) and res["dropin_autopatch"].get("value"):
                res["dropin_autopatch"]["posed_frames_identity_camera"] = posed_frames_leg(ts, lib, scene, settings, dev, steps=96, warmup=24)
        except Exception as e:
            res["dropin_autopatch"]["posed_frames_identity_camera"] = {"error": repr(e)}
        del params, den
        torch.cuda.empty_cache()
        extra = {}
        if (N, W, H) == (1_000_000, 980, 545) and not args.clustered:
            legs = [("C2 300k @980x545", dict(N=300_000, W=980, H=545, steps=20)),
                    ("C3 1M @1920x1080, densification every 100 steps", dict(N=1_000_000, W=1920, H=1080, steps=300, densify_every=100)),
                    ("C5 4M @980x545", dict(N=4_000_000, W=980, H=545, steps=10)),
                    ("syn-1M clustered @980x545", dict(N=1_000_000, W=980, H=545, steps=20, clustered=True)),
                    ("stage-A size 50k @980x545, SH degree 3", dict(N=50_000, W=980, H=545, steps=50)),
                    ("stage-A size 20k @980x545, SH degree 3", dict(N=20_000, W=980, H=545, steps=50)),
                    # what stage A actually runs: active degree 0 with 16 coefficients stored (gaussian_model_ht.py:68; no oneupSHdegree
                    # in train_single_image_3DGS / train_relative_pose)
                    ("stage-A size 130k @980x545, SH degree 0 (16 stored)", dict(N=130_000, W=980, H=545, steps=50, deg=0)),
                    ("stage-A size 50k @980x545, SH degree 0 (16 stored)", dict(N=50_000, W=980, H=545, steps=50, deg=0)),
                    ("stage-A size 20k @980x545, SH degree 0 (16 stored)", dict(N=20_000, W=980, H=545, steps=50, deg=0)),
                    ("C2 300k @980x545, SH degree 1 (16 stored)", dict(N=300_000, W=980, H=545, steps=20, deg=1))]
            for name, kw in legs:
                try:
                    extra[name] = workload_leg(syn, ts, raster, dev, warmup=3 if kw["N"] >= 1_000_000 else 10, **{"deg": deg, **kw})
                except Exception as e:
                    extra[name] = {"error": repr(e)}
            try:
                extra["stage-A frame pair @980x545 (single-image model + pose fit)"] = stage_a_leg(dev)
            except Exception as e:
                extra["stage-A frame pair @980x545 (single-image model + pose fit)"] = {"error": repr(e)}
            try:
                extra["stage-A image iteration, 8 pairs per launch chain (GsrBatch)"] = stage_a_batched_leg(dev)
            except Exception as e:
                extra["stage-A image iteration, 8 pairs per launch chain (GsrBatch)"] = {"error": repr(e)}
            try:
                extra["distCUDA2 (simple_knn._C), HIP"] = knn_leg(syn, dev)
            except Exception as e:
                extra["distCUDA2 (simple_knn._C), HIP"] = {"error": repr(e)}
        res["other_workloads"] = extra
    if world == 1 and not args.no_cpu_baseline:
        threads, quota = _CPUS, _CPU_QUOTA
        try:
            res["cpu_baseline"] = cpu_baseline(scene, threads)
            res["cpu_baseline"]["host"] = {"visible_cores": os.cpu_count(), "cpu_quota": quota,
                                           "note": "cores = OpenMP threads used = visible cores capped by the container's CPU-time quota"}
        except Exception as e:  # the checker must never take the bench line down
            res["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": threads, "kind": "port",
                                   "sample": f"failed: {e}"}
    state["printed"] = True
    print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
