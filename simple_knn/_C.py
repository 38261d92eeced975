"""`distCUDA2(points[N,3]) -> [N]`: mean squared distance to the 3 nearest neighbours, on the MI355X.

Drop-in for the reference's un-vendored `simple_knn._C` (/root/reference/.gitmodules:1-3), imported at
/root/reference/scene/gaussian_model_ht.py:20 and called at :211-216.  Semantics pinned by the reference's own
SciPy twin (:31-36: KDTree.query(k=4), drop self, mean of squared distances).  Exact 3-NN in hand-written HIP
(csrc/knn_kernels.hip: Morton sort + per-box AABB pruning) behind the C ABI `gsr_knn_mean_dist2`.
No CPU fallback: a non-device tensor raises.
"""
import ctypes as C
import importlib

import torch

_L = importlib.import_module("3dgs_hierarchical_training_amd._lib")


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    if points.device.type != "cuda":
        raise RuntimeError("distCUDA2: points must be on a ROCm/HIP device (no CPU fallback)")
    lib = _L.load()
    p = points.detach().float().contiguous()
    n = p.shape[0]
    out = torch.empty(n, dtype=torch.float32, device=p.device)
    if n == 0:
        return out
    sb = lib.gsr_knn_scratch_bytes(n)
    scratch = torch.empty(sb, dtype=torch.uint8, device=p.device)
    with torch.cuda.device(p.device):
        st = torch.cuda.current_stream(p.device).cuda_stream
        _L.check(lib.gsr_knn_mean_dist2(p.data_ptr(), n, out.data_ptr(), scratch.data_ptr(), sb, C.c_void_p(st)),
                 "gsr_knn_mean_dist2")
    return out
