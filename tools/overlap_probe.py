"""Feasibility probe: do the latency-bound radix sorts overlap with an HBM-bound streaming kernel when they run on two
streams?  (sorts: gsr_sort_pairs_u32 over 1 M keys + gsr_sort_pairs_u16 over 4.5 M keys; stream kernel: gsr_adam_step over
59 M floats = the multi-tensor Adam, 1.65 kB per Gaussian).  Prints serial vs concurrent wall time.
gpurun -- 'python tools/overlap_probe.py'"""
import ctypes as C, importlib, sys, time
import torch
sys.path.insert(0, '.')
L = importlib.import_module("3dgs_hierarchical_training_amd._lib")
lib = L.load()
dev = torch.device("cuda:0")
N, R = 1_000_000, 4_500_000
g = torch.Generator(device=dev).manual_seed(0)
k32 = torch.randint(0, 2 ** 31 - 1, (N,), device=dev, dtype=torch.int32, generator=g)
v32 = torch.arange(N, device=dev, dtype=torch.int32)
k32b, v32b = torch.empty_like(k32), torch.empty_like(v32)
k16 = torch.randint(0, 2170, (R,), device=dev, dtype=torch.int16, generator=g)
v16 = torch.arange(R, device=dev, dtype=torch.int32)
k16b, v16b = torch.empty_like(k16), torch.empty_like(v16)
sc32 = torch.empty(lib.gsr_sort_scratch_bytes(N), dtype=torch.uint8, device=dev)
sc16 = torch.empty(lib.gsr_sort_scratch_bytes(R), dtype=torch.uint8, device=dev)
p = torch.zeros(59 * N, device=dev); gr = torch.ones_like(p); m = torch.zeros_like(p); v = torch.zeros_like(p)
arr = (L.GsrAdamTensor * 1)()
arr[0].param, arr[0].grad, arr[0].exp_avg, arr[0].exp_avg_sq, arr[0].n, arr[0].lr = p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), 1e-3
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=-1)
flag = C.c_int(0)

def sorts(st):
    L.check(lib.gsr_sort_pairs_u32(k32.data_ptr(), v32.data_ptr(), k32b.data_ptr(), v32b.data_ptr(), N, 0, 32, sc32.data_ptr(), sc32.numel(), C.byref(flag), C.c_void_p(st.cuda_stream)), "sort32")
    L.check(lib.gsr_sort_pairs_u16(k16.data_ptr(), v16.data_ptr(), k16b.data_ptr(), v16b.data_ptr(), R, 0, 12, sc16.data_ptr(), sc16.numel(), C.byref(flag), C.c_void_p(st.cuda_stream)), "sort16")

def adam(st):
    L.check(lib.gsr_adam_step(arr, 1, 0.9, 0.999, 1e-15, 1, C.c_void_p(st.cuda_stream)), "adam")

def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / reps

t_sort = timeit(lambda: sorts(sa))
t_adam = timeit(lambda: adam(sa))
t_serial = timeit(lambda: (adam(sa), sorts(sa)))
def conc(first_adam=True, prio=False):
    s_sort = sb if prio else sa
    s_adam = sa if prio else sb
    def f():
        e = torch.cuda.Event(); e.record(sa)
        s_sort.wait_event(e) if s_sort is not sa else None
        s_adam.wait_event(e) if s_adam is not sa else None
        if first_adam:
            adam(s_adam); sorts(s_sort)
        else:
            sorts(s_sort); adam(s_adam)
        e2 = torch.cuda.Event(); e2.record(sb); sa.wait_event(e2)
    return f
print(f"sorts alone {t_sort:.0f} us, adam alone {t_adam:.0f} us, serial {t_serial:.0f} us")
for fa in (True, False):
    for pr in (False, True):
        print(f"concurrent (adam first={fa}, sorts on high-priority stream={pr}): {timeit(conc(fa, pr)):.0f} us")
