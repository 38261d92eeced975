import importlib, sys, ctypes as C, torch
sys.path.insert(0, ".")
import bench
L = importlib.import_module("3dgs_hierarchical_training_amd._lib")
lib = L.load()
dev = torch.device("cuda:0")
a = torch.empty(1 << 30, dtype=torch.uint8, device=dev); b = torch.empty_like(a); a.zero_()
st = torch.cuda.current_stream(dev).cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for variant in (0, 1, 2, 3):
    for blocks in ((256, 512, 1024, 2048) if variant == 3 else (2048, 4096, 8192, 16384)):
        for _ in range(2): lib.gsr_stream_copy(a.data_ptr(), b.data_ptr(), 1 << 30, variant, blocks, C.c_void_p(st))
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): lib.gsr_stream_copy(a.data_ptr(), b.data_ptr(), 1 << 30, variant, blocks, C.c_void_p(st))
        e1.record(); torch.cuda.synchronize()
        print(variant, blocks, "%.2f TB/s" % (2.0 * (1 << 30) / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e12))
print(bench.copy_ceiling(lib, dev))
