"""Stage-A pose fit on synthetic frame pairs under several settings (which of them make the photometric SE(3) fit of
stage_a.fit_pair land on the true relative pose).   gpurun -- 'python tools/stage_a_probe.py'"""
import importlib, sys, torch
sys.path.insert(0, '.')
sequence = importlib.import_module("3dgs_hierarchical_training_amd.sequence")
stage_a = importlib.import_module("3dgs_hierarchical_training_amd.stage_a")
pose = importlib.import_module("3dgs_hierarchical_training_amd.pose")
dev = torch.device("cuda:0")
def run(tag, F, gtn, W, H, seed, ang, shift, n, ii, pi, lr, pairs):
    seq = sequence.FrameSequence(F, gtn, W, H, dev, seed=seed, step_angle=ang, step_shift=shift)
    out = []
    for p in pairs:
        M = stage_a.fit_pair(seq, p, dev, n_points=n, single_image_iters=ii, pose_iters=pi, pose_lr=lr, seed=0)
        T = seq.true_rel_pose(p, p + 1)
        out.append((p, "t_err %.4f of %.4f" % (float((M[:3, 3] - T[:3, 3]).norm()), float(T[:3, 3].norm())),
                    "R_err %.4f of %.4f" % (float((M[:3, :3] - T[:3, :3]).abs().max()), float((torch.eye(3) - T[:3, :3]).abs().max()))))
    print(tag, out, flush=True)
run("test-config      ", 6, 5000, 256, 192, 2, 0.015, 0.02, 5000, 150, 250, 1e-3, [2])
run("small angle      ", 6, 5000, 256, 192, 2, 0.006, 0.02, 5000, 150, 250, 1e-3, [2])
run("seed 0           ", 6, 5000, 256, 192, 0, 0.006, 0.02, 5000, 150, 250, 1e-3, [0, 2])
run("60k gt, 20k model", 6, 60000, 320, 240, 0, 0.006, 0.02, 20000, 150, 250, 1e-3, [0, 2])
run("60k gt, 60k model", 6, 60000, 320, 240, 0, 0.006, 0.02, 60000, 150, 250, 1e-3, [0, 2])
run("no image training", 6, 60000, 320, 240, 0, 0.006, 0.02, 60000, 0, 250, 1e-3, [0, 2])
