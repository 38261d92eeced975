"""Do two independent small-model train loops on two streams (two host threads, one process) overlap on the device?
gpurun -- 'python tools/two_streams_probe.py'"""
import importlib, sys, threading, time, torch
sys.path.insert(0, '.')
sequence = importlib.import_module("3dgs_hierarchical_training_amd.sequence")
ts = importlib.import_module("3dgs_hierarchical_training_amd.train_step")
host = importlib.import_module("3dgs_hierarchical_training_amd.host"); host.cap_host_threads()
dev = torch.device("cuda:0")
seq = sequence.FrameSequence(4, 400000, 980, 545, dev, seed=0)
ident = seq.settings_for_pose(torch.eye(4))
models = []
for f in range(4):
    sc = seq.pixel_scene(f % 3, stride=2, seed=f)
    models.append((ts.GaussianParams(sc, dev), seq.target(f % 3)))
def loop(k, n, stream):
    p, tgt = models[k]
    with torch.cuda.stream(stream):
        for _ in range(n):
            ts.train_step(p, ident, tgt, next_settings=ident)
for k in range(4):
    loop(k, 30, torch.cuda.current_stream())
torch.cuda.synchronize()
n = 600
t0 = time.perf_counter(); loop(0, n, torch.cuda.current_stream()); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"one loop: {1e3 * (t1 - t0) / n:.4f} ms per step")
for T in (2, 3, 4):
    streams = [torch.cuda.Stream(dev) for _ in range(T)]
    for s_ in streams: s_.wait_stream(torch.cuda.current_stream())
    th = [threading.Thread(target=loop, args=(k, n, streams[k])) for k in range(T)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"{T} loops on {T} streams / threads: {1e3 * (t1 - t0) / (T * n):.4f} ms per step")
