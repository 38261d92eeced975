"""Host-side housekeeping of the application entry points (run_segments.py; bench.py carries its own copy because it has to
act before `import torch`).

`cap_host_threads()` sizes torch's CPU thread pool to what the container may actually use: min(visible cores, cgroup CPU quota).
On the MI355X boxes 256 cores are visible under a 16-CPU quota; an OpenMP team of 256 spinning after a CPU-side op exhausts the
cgroup's CPU time and the kernel throttles every thread of the process, the kernel-launching one included (INTEGRATION.md,
"Deployment note")."""
import os


def usable_cpus():
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


def cap_host_threads():
    """Returns the thread count now in effect.  An explicit OMP_NUM_THREADS of the caller wins (torchrun sets 1)."""
    import torch
    if "OMP_NUM_THREADS" not in os.environ:
        n, _ = usable_cpus()
        torch.set_num_threads(min(torch.get_num_threads(), n))
    return torch.get_num_threads()
