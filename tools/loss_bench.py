"""Times the fused photometric loss kernels alone (forward, backward) at 3 x 545 x 980 with HIP events.
gpurun -- 'python tools/loss_bench.py'"""
import importlib, sys
import torch
sys.path.insert(0, '.')
loss_mod = importlib.import_module("3dgs_hierarchical_training_amd.loss")
dev = torch.device("cuda:0")
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (545, 980)
g = torch.Generator().manual_seed(0)
gt = torch.rand(3, H, W, generator=g).to(dev)
raw = (gt.cpu() + 0.3 * torch.randn(3, H, W, generator=g)).to(dev).requires_grad_(True)
def run(n, bwd):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = loss_mod.fused_photometric_loss(raw, gt, 0.2, clamp=True)
        if bwd:
            out.backward(); raw.grad = None
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
run(20, True)
f = run(300, False); fb = run(300, True)
print(f"loss forward {f:.1f} us, forward+backward {fb:.1f} us per call ({H}x{W})")
