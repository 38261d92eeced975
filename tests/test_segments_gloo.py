"""CPU, world_size 2 over gloo: the merge-step exchange of the one-segment-per-GPU sharding."""
import importlib
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_segment(n, seed):
    g = torch.Generator().manual_seed(seed)
    return {"_xyz": torch.randn(n, 3, generator=g), "_features_dc": torch.randn(n, 1, 3, generator=g),
            "_features_rest": torch.randn(n, 15, 3, generator=g), "_opacity": torch.randn(n, 1, generator=g),
            "_scaling": torch.randn(n, 3, generator=g), "_rotation": torch.randn(n, 4, generator=g)}


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    seg_mod = importlib.import_module("3dgs_hierarchical_training_amd.segments")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = _make_segment(100 + 37 * rank, seed=rank)
    pairs = seg_mod.merge_schedule(world)[0]
    role = seg_mod.partner(rank, pairs)
    pose = torch.arange(7, dtype=torch.float32) + rank
    ok = True
    if role[0] == "send":
        seg_mod.send_segment(mine, role[1], extra=pose)
    else:
        got, extra = seg_mod.recv_segment(role[1], torch.device("cpu"))
        expect = _make_segment(100 + 37 * role[1], seed=role[1])
        ok = all(torch.equal(got[k], expect[k]) for k in seg_mod.SEGMENT_KEYS) and torch.equal(extra, torch.arange(7.0) + role[1])
        keep_d = torch.rand(mine["_xyz"].shape[0]) > 0.5
        keep_s = torch.rand(got["_xyz"].shape[0]) > 0.5
        T = torch.eye(4); T[:3, 3] = torch.tensor([1.0, 2.0, 3.0])
        merged = seg_mod.merge_segments(mine, got, keep_d, keep_s, T)
        n = int(keep_d.sum() + keep_s.sum())
        ok = ok and all(merged[k].shape[0] == n for k in seg_mod.SEGMENT_KEYS)
        ok = ok and torch.allclose(merged["_xyz"][int(keep_d.sum()):], got["_xyz"][keep_s] + torch.tensor([1.0, 2.0, 3.0]))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


def test_merge_exchange_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def _merge_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hier = importlib.import_module("3dgs_hierarchical_training_amd.hierarchy")
    seg_mod = importlib.import_module("3dgs_hierarchical_training_amd.segments")
    seg = _make_segment(200 + 10 * rank, seed=rank)

    def fake_importance(s, views):          # stands in for the HIP rasterizer on the CPU test box
        return s["_xyz"].abs().sum(1, keepdim=True).repeat(1, 48)

    n0 = seg["_xyz"].shape[0]
    for pairs in seg_mod.merge_schedule(world):
        if seg is None:
            break
        seg = hier.merge_level(seg, [], pairs, 0.5, importance_fn=fake_importance)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, None if seg is None else int(seg["_xyz"].shape[0]), n0))


def test_merge_tree_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_merge_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    (r0, n_merged, n0), (r1, none1, n1) = res
    assert none1 is None                                   # the source rank handed its segment over
    assert n_merged == (n0 - n0 // 2) + (n1 - n1 // 2)      # both sides pruned by prune_ratio = 0.5


def test_merge_schedule_tree():
    seg_mod = importlib.import_module("3dgs_hierarchical_training_amd.segments")
    lv = seg_mod.merge_schedule(8)
    assert lv == [[(0, 1), (2, 3), (4, 5), (6, 7)], [(0, 2), (4, 6)], [(0, 4)]]
    assert seg_mod.partner(3, lv[0]) == ("send", 2) and seg_mod.partner(4, lv[2]) == ("send", 0)
    assert seg_mod.partner(1, lv[1]) is None


# ---- config 4 tree walk (run_segments.RankRunner) over gloo, world 2 and 4 ----------------------------------------------
# The three places that reach the HIP rasterizer are replaced by CPU stand-ins; everything else -- partition, pose chains,
# un-pruned child + mask messages, masks applied at the destination, teachers, frame unions -- is the product code.
def _fake_importance(s, views):
    return s["_xyz"].abs().sum(1, keepdim=True).repeat(1, 48) + 0.01 * len(views)


def _tree_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rs = importlib.import_module("3dgs_hierarchical_training_amd.run_segments")
    seg_mod = importlib.import_module("3dgs_hierarchical_training_amd.segments")
    sequence = importlib.import_module("3dgs_hierarchical_training_amd.sequence")

    class Seq(sequence.FrameSequence):
        def target(self, f):
            return torch.zeros(3, self.H, self.W)

    cfg = rs.HTConfig(frames=24, width=64, height=48, gt_gaussians=600, leaf_gaussians=200 + 10 * rank, leaf_iters_per_frame=2,
                      phase1_iters_per_frame=2, phase2_iters_per_frame=[1, 1, 1], optimizer="torch", fused=False)
    seq = Seq(cfg.frames, cfg.gt_gaussians, cfg.width, cfg.height, torch.device("cpu"))
    log, steps, renders = [], [0], [0]

    def step_fn(seg, settings, target):
        steps[0] += 1

    def teacher_render(seg, settings):
        renders[0] += 1
        return torch.zeros(3, cfg.height, cfg.width)

    rr = rs.RankRunner(rank, world, seg_mod.DistTransport(), seq, cfg, torch.device("cpu"), log=log.append,
                       importance_fn=_fake_importance, step_fn=step_fn, teacher_render_fn=teacher_render)
    def all_ok(mine):                     # the combiner run_segments.py passes: one MIN all-reduce of the ranks' self-test verdicts
        flag = torch.tensor([1.0 if mine else 0.0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item() >= 1.0)
    final = rr.run(barrier=dist.barrier, all_ok=all_ok)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, None if final is None else (final.params.num_points, final.frames, final.start_fidx, sorted(final.poses)),
           log, steps[0], renders[0]))


def _run_tree(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tree_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    return res


def _check_tree(world):
    res = _run_tree(world)
    n_leaf = [200 + 10 * r for r in range(world)]
    # expected Gaussian counts up the tree: both children lose int(N * 0.5) at every merge
    counts, step = list(n_leaf), 1
    while step < world:
        for k in range(0, world, 2 * step):
            counts[k] = (counts[k] - counts[k] // 2) + (counts[k + step] - counts[k + step] // 2)
        step *= 2
    rank0 = res[0]
    assert rank0[1][0] == counts[0]
    assert rank0[1][1] == list(range(24)) and rank0[1][2] == 0 and rank0[1][3] == list(range(24))   # all frames, all poses chained
    for rank, final, log, steps, renders in res[1:]:
        assert final is None                           # every other rank handed its model up the tree
    levels = world.bit_length() - 1
    for rank, final, log, steps, renders in res:
        # before any training every rank tested each edge of the merge tree it takes part in (1 MB there and back)
        st = [r for r in log if r["phase"] == "link_selftest"]
        assert len(st) == 1 and st[0]["ok"] and log[0]["phase"] == "link_selftest"
        n_edges = levels if rank == 0 else (rank & -rank).bit_length()      # rank r merges at levels 0 .. trailing-zeros(r)
        assert len(st[0]["pairs"]) == n_edges and all(p["ok"] for p in st[0]["pairs"])
        merges = [r for r in log if r["phase"] == "merge"]
        for m in merges:
            if m["role"] == "src":                     # the child travels UN-pruned: 59 floats per Gaussian + 1 mask byte
                assert m["bytes"] >= m["n"] * (59 * 4 + 1) and m["n_dropped"] == m["n"] // 2
            else:
                assert m["bytes"] >= m["n_child"] * (59 * 4 + 1)
                assert m["n_merged"] == (m["n"] - m["n_dropped"]) + (m["n_child"] - m["n_child_dropped"])
                assert m["n_child_dropped"] == m["n_child"] // 2
        non = [r for r in log if r["phase"] == "nonleaf"]
        assert len(non) == sum(1 for m in merges if m["role"] == "dst")
        if non:
            assert renders == sum(r["virtual_views"] for r in non) and renders > 0    # phase 1 asked the frozen children
            assert steps > 0


def test_segment_tree_world2():
    _check_tree(2)


def test_segment_tree_world4():
    _check_tree(4)


def test_local_transport_walks_the_same_tree():
    """One process, LocalTransport: the same RankRunner code yields the same counts as the distributed walk."""
    rs = importlib.import_module("3dgs_hierarchical_training_amd.run_segments")
    sequence = importlib.import_module("3dgs_hierarchical_training_amd.sequence")

    class Seq(sequence.FrameSequence):
        def target(self, f):
            return torch.zeros(3, self.H, self.W)

    cfg = rs.HTConfig(frames=40, width=64, height=48, gt_gaussians=600, leaf_gaussians=300, leaf_iters_per_frame=1,
                      phase1_iters_per_frame=1, phase2_iters_per_frame=[1], optimizer="torch", fused=False)
    seq = Seq(cfg.frames, cfg.gt_gaussians, cfg.width, cfg.height, torch.device("cpu"))
    seg_mod = importlib.import_module("3dgs_hierarchical_training_amd.segments")
    tr = seg_mod.LocalTransport(8)
    runners = [rs.RankRunner(r, 8, tr, seq, cfg, torch.device("cpu"), log=lambda rec: None, importance_fn=_fake_importance,
                             step_fn=lambda *a: None, teacher_render_fn=lambda s, st: torch.zeros(3, 48, 64)) for r in range(8)]
    for rr in runners:
        rr.train_leaf()
    for k, pairs in enumerate(runners[0].schedule):
        for dst, src in pairs:
            runners[src].merge_send(k)
        for dst, src in pairs:
            runners[dst].merge_recv(k)
            # the destination holds both UN-pruned children as teachers
            t_own, t_child = runners[dst].teachers
            msgs = [r for r in runners[dst].report if r["phase"] == "merge" and r["level"] == k]
            assert t_own["seg"]["_xyz"].shape[0] == msgs[-1]["n"] and t_child["seg"]["_xyz"].shape[0] == msgs[-1]["n_child"]
            runners[dst].train_nonleaf(k)
    assert runners[0].seg.frames == list(range(40))
    assert all(r.seg is None for r in runners[1:])
    n = 300
    for _ in range(3):
        n = 2 * (n - n // 2)
    assert runners[0].seg.params.num_points == n


def _stage_a_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sa = importlib.import_module("3dgs_hierarchical_training_amd.stage_a")
    fitted = []

    def fit(p):            # stands in for stage_a.fit_pair (HIP): a pose that encodes the pair index
        fitted.append(p)
        M = torch.eye(4)
        M[:3, 3] = torch.tensor([p, 10.0 * p, -p])
        return M

    d = sa.run_stage_a(11, fit, torch.device("cpu"))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, fitted, {k: v[:3, 3].tolist() for k, v in d.items()}))


def test_stage_a_round_robin_all_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stage_a_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res[0][1] == [0, 2, 4, 6, 8] and res[1][1] == [1, 3, 5, 7, 9]          # pairs dealt round-robin
    for rank, fitted, d in res:                                                     # every rank ends with the whole table
        assert sorted(d) == sorted(f"rel_pose_{p}_to_{p + 1}" for p in range(10))
        for p in range(10):
            assert d[f"rel_pose_{p}_to_{p + 1}"] == [float(p), 10.0 * p, float(-p)]


def _self_worker(rank, world, port, q):
    """Rank 1 of a two-rank group addresses a whole child message (header, rows, mask, frames, poses) to ITSELF while rank 0 looks
    on: segments.DistTransport holds a self-addressed send back and pairs it with the receive in one grouped call."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    seg_mod = importlib.import_module("3dgs_hierarchical_training_amd.segments")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok, err = True, ""
    try:
        if rank == world - 1:
            tr = seg_mod.DistTransport()
            mine = _make_segment(321, seed=5)
            drop = torch.rand(321) > 0.5
            poses = torch.randn(3, 4, 4)
            st = seg_mod.send_child(tr, rank, mine, drop=drop, frames=[4, 5, 6], poses=poses, start_fidx=4, global_iteration=77, sh_degree=2)
            msg = seg_mod.recv_child(tr, rank, torch.device("cpu"))
            ok = all(torch.equal(msg["seg"][k], mine[k]) for k in seg_mod.SEGMENT_KEYS) and torch.equal(msg["drop"], drop) and \
                msg["frames"] == [4, 5, 6] and torch.equal(msg["poses"], poses) and msg["start_fidx"] == 4 and \
                msg["global_iteration"] == 77 and msg["sh_degree"] == 2 and msg["bytes"] == st["bytes"]
            try:
                tr.recv(torch.zeros(3), rank)
                ok = False
            except RuntimeError as e:
                ok = ok and "before it sent" in str(e)
    except Exception as e:    # noqa: BLE001
        ok, err = False, repr(e)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok), err))


def test_a_rank_can_address_a_child_message_to_itself():
    for world in (1, 2):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_self_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=120) for _ in procs]
        for p in procs:
            p.join(timeout=60)
        assert all(r[1] for r in res), res
