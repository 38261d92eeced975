"""`distCUDA2(points[N,3]) -> [N]`: mean squared distance to the 3 nearest neighbours, on the MI355X.

Drop-in for the reference's un-vendored `simple_knn._C` (/root/reference/.gitmodules:1-3), imported at
/root/reference/scene/gaussian_model_ht.py:20 and called at :211-216.  Semantics pinned by the reference's own
SciPy twin (:31-36: KDTree.query(k=4), drop self, mean of squared distances).  Exact 3-NN in hand-written HIP
(csrc/knn_kernels.hip: Morton sort + per-box AABB pruning) behind the C ABI `gsr_knn_mean_dist2`, reached through the PyTorch
extension op `torch.ops.gsr.knn_mean_dist2` (csrc/torch_ext.cpp).
No CPU fallback: a non-device tensor raises.
"""
import importlib

import torch

_E = importlib.import_module("3dgs_hierarchical_training_amd._ext")


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    if points.device.type != "cuda":
        raise RuntimeError("distCUDA2: points must be on a ROCm/HIP device (no CPU fallback)")
    return _E.load().knn_mean_dist2(points.detach())
