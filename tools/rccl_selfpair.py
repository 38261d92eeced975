#!/usr/bin/env python3
"""Does RCCL accept a point-to-point SELF pair at world 1?  (`dist.send / recv` on device tensors has never
executed because a gpurun box has one GPU.)  Tries, each under its own watchdog so a hang is reported instead of waited out:
  a) batch_isend_irecv([isend(t, 0), irecv(r, 0)])   -- one grouped call, the NCCL-documented way to talk to oneself
  b) the same with 64 MiB
Prints one JSON line per attempt."""
import json
import os
import sys
import threading
import time

import torch
import torch.distributed as dist


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    for name, n in (("self_pair_1MiB", 1 << 18), ("self_pair_64MiB", 1 << 24)):
        res = {"attempt": name, "ok": None}
        done = threading.Event()

        def dog():
            if not done.wait(60.0):
                print(json.dumps({"attempt": name, "ok": False, "error": "no completion within 60 s (watchdog)"}), flush=True)
                os._exit(4)
        threading.Thread(target=dog, daemon=True).start()
        try:
            a = torch.arange(n, dtype=torch.float32, device=dev)
            b = torch.zeros_like(a)
            torch.cuda.synchronize(dev)
            for rep in range(3):
                t0 = time.perf_counter()
                works = dist.batch_isend_irecv([dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.irecv, b, 0)])
                for w in works:
                    w.wait()
                torch.cuda.synchronize(dev)
                dt = time.perf_counter() - t0
            res.update(ok=bool(torch.equal(a, b)), bytes=4 * n, ms=1e3 * dt, GBps=4 * n / dt / 1e9)
        except Exception as e:
            res.update(ok=False, error=repr(e)[:500])
        done.set()
        print(json.dumps(res), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
