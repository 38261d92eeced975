"""Feasibility probe (round 5): a persistent streaming kernel with a bounded footprint per CU (tools/microbench/persist_stream.hip:
B blocks of 256 threads per CU, no LDS, U 16-byte elements per stream in flight per thread) on one stream, the library's whole
forward (preprocess + depth sort + direct binning + blend) on another.  Does the latency-bound forward find room next to it?
Prints each alone, back to back, and together.        gpurun -- 'python tools/overlap_persist.py'"""
import ctypes as C, importlib, os, subprocess, sys, time
import torch
sys.path.insert(0, '.')
HERE = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/persist_stream.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                       os.path.join(HERE, "microbench", "persist_stream.hip"), "-o", so])
ps = C.CDLL(so)
ps.persist_stream.argtypes = [C.c_void_p] * 6 + [C.c_size_t, C.c_int, C.c_int, C.c_void_p]
syn = importlib.import_module('3dgs_hierarchical_training_amd.synthetic')
ts = importlib.import_module('3dgs_hierarchical_training_amd.train_step')
host = importlib.import_module("3dgs_hierarchical_training_amd.host"); host.cap_host_threads()
from diff_gaussian_rasterization import GaussianRasterizer
dev = torch.device('cuda:0')
N, W, H = 1_000_000, 980, 545
scene = syn.make_scene(N, W, H, sh_degree=3, seed=3)
rs = ts.make_settings(scene, dev, 3)
t = {k: scene[k].to(dev) for k in ["means3D", "shs", "opacities", "scales", "rotations"]}
m2d = torch.zeros(N, 3, device=dev)
n4 = 48 * N // 4          # the f_dc + f_rest groups: 48 floats per Gaussian, parameter + two moments in and out = 1 152 B
bufs = [torch.rand(n4 * 4, device=dev) for _ in range(3)]
outs = [torch.empty(n4 * 4, device=dev) for _ in range(3)]
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
CUS = torch.cuda.get_device_properties(0).multi_processor_count

def fwd():
    with torch.no_grad(), torch.cuda.stream(sa):
        GaussianRasterizer(rs)(means3D=t["means3D"], means2D=m2d, shs=t["shs"], colors_precomp=None, opacities=t["opacities"],
                               scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)

def stream(blocks, unroll, st):
    rc = ps.persist_stream(*[b.data_ptr() for b in bufs], *[o.data_ptr() for o in outs], n4, blocks, unroll, C.c_void_p(st.cuda_stream))
    assert rc == 0, rc

def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / reps

t_f = timeit(fwd)
print(f"CUs {CUS}; forward alone {t_f:.0f} us; stream bytes {6 * 16 * n4 / 1e6:.0f} MB", flush=True)
for bpc, u in ((8, 1), (8, 2), (4, 2), (4, 4), (2, 4), (2, 8), (1, 8), (3, 4)):
    blocks = CUS * bpc
    t_s = timeit(lambda: stream(blocks, u, sb))
    def serial():
        fwd(); stream(blocks, u, sa)
    def both():
        e = torch.cuda.Event(); e.record(sa); sb.wait_event(e)
        stream(blocks, u, sb); fwd()
        e2 = torch.cuda.Event(); e2.record(sb); sa.wait_event(e2)
    def both2():
        e = torch.cuda.Event(); e.record(sa); sb.wait_event(e)
        fwd(); stream(blocks, u, sb)
        e2 = torch.cuda.Event(); e2.record(sb); sa.wait_event(e2)
    print(f"{bpc} blocks/CU x unroll {u}: stream alone {t_s:.0f} us = {6 * 16 * n4 / t_s / 1e6:.2f} TB/s; back to back {timeit(serial):.0f}; "
          f"together (stream first) {timeit(both):.0f}, (forward first) {timeit(both2):.0f} us", flush=True)
