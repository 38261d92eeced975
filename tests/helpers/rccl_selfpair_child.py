#!/usr/bin/env python3
"""One rank, backend nccl (= RCCL): a whole child message of the merge step (header, un-pruned 59-float rows, drop mask, frames,
poses) travels through segments.send_child / recv_child on DEVICE tensors, addressed to the rank itself -- RCCL's send / recv
kernels execute as a grouped self pair.  Prints one JSON line.  Run by tests/test_gpu_segments.py in a subprocess."""
import importlib
import json
import os
import sys
import threading

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    seg_mod = importlib.import_module("3dgs_hierarchical_training_amd.segments")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    done = threading.Event()

    def dog():
        if not done.wait(120.0):
            print(json.dumps({"ok": False, "error": "no completion within 120 s"}), flush=True)
            os._exit(4)
    threading.Thread(target=dog, daemon=True).start()
    g = torch.Generator().manual_seed(3)
    n = 200_000
    seg = {"_xyz": torch.randn(n, 3, generator=g), "_features_dc": torch.randn(n, 1, 3, generator=g),
           "_features_rest": torch.randn(n, 15, 3, generator=g), "_opacity": torch.randn(n, 1, generator=g),
           "_scaling": torch.randn(n, 3, generator=g), "_rotation": torch.randn(n, 4, generator=g)}
    seg = {k: v.to(dev) for k, v in seg.items()}
    drop = (torch.rand(n, generator=g) > 0.5).to(dev)
    poses = torch.randn(5, 4, 4, generator=g).to(dev)
    tr = seg_mod.DistTransport()
    st = seg_mod.link_selftest(tr, [[(0, 0)]], dev) if False else None      # (a pair needs two ranks; the self pair is the message below)
    s = seg_mod.send_child(tr, 0, seg, drop=drop, frames=[7, 8, 9, 10, 11], poses=poses, start_fidx=7, global_iteration=1234, sh_degree=2)
    m = seg_mod.recv_child(tr, 0, dev)
    ok = all(torch.equal(m["seg"][k], seg[k]) for k in seg_mod.SEGMENT_KEYS) and torch.equal(m["drop"], drop) and \
        m["frames"] == [7, 8, 9, 10, 11] and torch.equal(m["poses"], poses) and m["start_fidx"] == 7 and m["global_iteration"] == 1234 and \
        m["sh_degree"] == 2 and all(v.is_cuda for v in m["seg"].values())
    done.set()
    print(json.dumps({"ok": bool(ok), "backend": dist.get_backend(), "bytes": s["bytes"], "recv_bytes": m["bytes"], "over_wire": tr._self_over_wire,
                      "send_ms": s["ms"], "recv_ms": m["ms"]}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
