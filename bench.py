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

Rank 0 prints ONE JSON line.  `roofline` describes the per-tile forward blend kernel (the kernel
BASELINE.json's north_star names): achieved = algorithmic bytes (44 R_eff + 28 P + 8 T) /
average launch duration measured with HIP events on the launch stream inside the timed region.
`cpu_baseline` times oracle/ (the CPU restatement, kind "port") on the host cores -- reported beside the GPU
number, never part of it.
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

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=980)
    ap.add_argument("--height", type=int, default=545)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    """oracle/ restatement (binary32 build) on the host cores: one full forward (K1-K6) of the same workload,
    plus the K1-K5 (preprocess + duplicate + sort + ranges) leg on its own."""
    from oracle import binding
    o = binding.OracleRender(means3D=scene["means3D"], opacities=scene["opacities"], viewmatrix=scene["viewmatrix"],
                             projmatrix=scene["projmatrix"], campos=scene["campos"], bg=scene["bg"],
                             image_height=scene["image_height"], image_width=scene["image_width"],
                             tanfovx=scene["tanfovx"], tanfovy=scene["tanfovy"], sh_degree=scene["sh_degree"],
                             shs=scene["shs"], scales=scene["scales"], rotations=scene["rotations"], precision="f32")
    # bounded sample: repeat the forward until ~10 s of wall time (all host threads busy) have been spent
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
    return {"value": 1.0 / t_full, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"{reps} forward renders (K1-K6: preprocess+duplicate+sort+ranges+blend; no backward / loss / Adam) of the "
                      f"same syn workload, R={R} before exact tile culling, OpenMP over {threads} host threads, binary32 "
                      f"oracle/gsr_oracle.c; about {reps * (t_pre + t_full):.0f} s of wall time",
            "k1_k5_preprocess_sort_images_per_s": 1.0 / t_pre, "k1_k5_seconds": t_pre, "k1_k6_seconds": t_full}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (no CPU fallback in the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("GSR_BENCH_FORCE_DIST") == "1":   # (the env var exercises the RCCL path on one GPU)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    syn = importlib.import_module("3dgs_hierarchical_training_amd.synthetic")
    ts = importlib.import_module("3dgs_hierarchical_training_amd.train_step")
    L = importlib.import_module("3dgs_hierarchical_training_amd._lib")
    lib = L.load()
    if args.fwd_ppt:
        lib.gsr_set_option(b"blend_fwd_ppt", args.fwd_ppt)
    if args.bwd_ppt:
        lib.gsr_set_option(b"blend_bwd_ppt", args.bwd_ppt)
    if args.sort_algo >= 0:
        lib.gsr_set_option(b"sort_algo", args.sort_algo)
    if args.tile_map >= 0:
        lib.gsr_set_option(b"tile_map", args.tile_map)

    N, W, H, deg = args.gaussians, args.width, args.height, args.sh_degree
    scene = syn.make_scene(N, W, H, sh_degree=deg, seed=rank)
    gt = syn.target_image(W, H, seed=1).to(dev)
    params = ts.GaussianParams(scene, dev)
    settings = ts.make_settings(scene, dev, deg)
    names = ["preprocess_fwd", "sort_depth", "scan", "emit", "sort_tile", "ranges", "blend_fwd", "blend_bwd",
             "preprocess_bwd"]

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        ts.train_step(params, settings, gt)
    # instance statistics from one un-timed forward (they do not change the timed work)
    with torch.no_grad():
        pkg = ts.render(params, settings)
    n_visible = int((pkg["radii"] > 0).sum().item())
    del pkg
    # timed region: HIP events only around the forward blend kernel (the roofline kernel); an event pair is a ~10 us
    # stream bubble, so the per-stage breakdown is taken from a few extra UNTIMED steps afterwards
    lib.gsr_set_option(b"profile", 2)
    read_profile(lib, names)  # drop anything recorded so far
    sync_all()
    t0 = time.perf_counter()
    stamps = [t0]
    for _ in range(args.steps):
        ts.train_step(params, settings, gt)
        stamps.append(time.perf_counter())   # host clock only (each step already waits for the forward's instance count)
    sync_all()
    elapsed = time.perf_counter() - t0
    lib.gsr_set_option(b"profile", 0)
    prof_blend = read_profile(lib, ["blend_fwd"])["blend_fwd"]
    lib.gsr_set_option(b"profile", 1)
    for _ in range(min(5, args.steps)):
        ts.train_step(params, settings, gt)
    torch.cuda.synchronize(dev)
    lib.gsr_set_option(b"profile", 0)
    prof = read_profile(lib, names)
    prof["blend_fwd"] = prof_blend   # the figure the roofline uses: measured inside the timed region
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # R and R_eff of the final state (one extra forward outside the timed region)
    raster = importlib.import_module("3dgs_hierarchical_training_amd.rasterizer")
    with torch.no_grad():
        ts.render(params, settings)
    info = raster.last_call_info()
    R, R_eff = info["num_rendered"], info["staged"]

    if rank != 0:
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
    bwd_ms = sum(stage_ms[k] or 0.0 for k in bwd_stages)
    blend_ms = stage_ms["blend_fwd"]
    alg_bytes = 44.0 * R_eff + 28.0 * P + 8.0 * T
    achieved = (alg_bytes / (blend_ms * 1e-3) / 1e9) if blend_ms else None
    # HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, gfx950
    # read-side x2 correction; tools/pmc_profile.sh + tools/pmc_summary.py).  PMC collection cannot run inside this
    # process, so the figure is taken from the committed summary when it was measured on this very workload.
    traffic = None
    pmc_file = os.path.join(REPO, "profiles", "r01_pmc_blend.json")
    if (N, W, H, deg) == (1_000_000, 980, 545, 3) and os.path.exists(pmc_file):
        try:
            ks = json.load(open(pmc_file))["kernels"]
            traffic = next(v["hbm_traffic_bytes"] for k, v in ks.items() if "k_blend_fwd" in k)
        except Exception:
            traffic = None
    V = n_visible
    bwd_ms, preb_ms = stage_ms["blend_bwd"], stage_ms["preprocess_bwd"]
    others = []
    if bwd_ms:
        ab = 44.0 * R_eff + 36.0 * P + 40.0 * V
        others.append({"kernel": "k_blend_bwd2", "bound": "hbm", "achieved": ab / (bwd_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": ab / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": ab,
                       "avg_launch_ms": bwd_ms, "note": "VALU-bound (packed f32 math + cross-lane reduction), see DESIGN.md"})
    if preb_ms:
        ab = 1476.0 * N   # params 236 + ggrad 48 + moments 472 in; params + moments 708 + means2D grad 12 out
        others.append({"kernel": "k_preprocess_bwd (per-Gaussian backward + in-kernel Adam)", "bound": "hbm",
                       "achieved": ab / (preb_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": ab / (preb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": ab,
                       "avg_launch_ms": preb_ms, "note": "durations from the untimed stage-profiling steps"})
    roofline = {"kernel": "k_blend_fwd_w", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                "traffic_source": "profiles/r01_pmc_blend.json (rocprofv3 --pmc, separate passes)" if traffic else None,
                "traffic_note": "81 MB with one band of tiles per XCD; dealing 2x2 tile blocks round-robin to the XCDs (load "
                                "balance) lets neighbouring L2s fetch some splat records twice (+~25 MB of reads), and the "
                                "per-pixel checkpoints for the split backward add ~25 MB of writes" if traffic else None,
                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": blend_ms,
                "launches_timed": prof["blend_fwd"][1], "R": R, "R_eff": R_eff, "P": P, "T": T}
    res = {
        "metric": "train-step images/sec + fwd+bwd ms @1M Gaussians, 980x545; 1/2/4/8 GPU",   # BASELINE.json's metric
        "value": world * args.steps / elapsed, "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"syn-{N} Gaussians, {W}x{H}, SH degree {deg}, identity-pose pinhole camera "
                               f"(FoVx {syn.FOVX_FRANCIS}), U[0,1] target, one view per step per GPU",
                   "gaussians": N, "width": W, "height": H, "sh_degree": deg, "visible": n_visible,
                   "num_rendered_R": R, "parallelism": f"{world} independent segment replica(s), no data-path collective",
                   "train_step": "activations + rasterize fwd + 0.8*L1+0.2*(1-SSIM) + backward + Adam(eps=1e-15) on all 59 floats "
                                 "per Gaussian (update applied inside the per-Gaussian backward kernel)"},
        "fwd_bwd_ms": fwd_ms + bwd_ms, "rasterizer_fwd_ms": fwd_ms, "rasterizer_bwd_ms": bwd_ms,
        "stage_ms": stage_ms, "step_host_ms": step_host, "roofline": roofline, "roofline_other_kernels": others,
    }
    if world == 1 and not args.no_cpu_baseline:
        threads = os.cpu_count() or 1
        try:
            res["cpu_baseline"] = cpu_baseline(scene, threads)
        except Exception as e:  # the checker must never take the bench line down
            res["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": threads, "kind": "port",
                                   "sample": f"failed: {e}"}
    print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
