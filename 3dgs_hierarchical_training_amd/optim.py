"""Multi-tensor Adam on top of gsr_adam_step (one HIP launch for all parameter groups).

Mirror of how the reference drives its optimizer: `torch.optim.Adam(l, lr=0.0, eps=1e-15)` with one group
per parameter tensor and per-group learning rates (/root/reference/scene/gaussian_model_ht.py:275-289);
`step()` / `zero_grad(set_to_none=True)` as called at /root/reference/trainer/ht3dgs_trainer.py:159-166.
Same update rule (no amsgrad, no weight decay).  No CPU path.
"""
import ctypes as C
from typing import Dict, List

import torch

from . import _lib as L


class FusedAdam:
    def __init__(self, param_groups: List[Dict], lr: float = 0.0, betas=(0.9, 0.999), eps: float = 1e-15):
        self.param_groups = []
        for g in param_groups:
            g = dict(g)
            g.setdefault("lr", lr)
            assert len(g["params"]) == 1, "one tensor per group, as the reference builds them"
            self.param_groups.append(g)
        n = sum(1 for _ in self.param_groups)
        assert n <= 8, "gsr_adam_step handles up to 8 tensors per launch"
        self.betas, self.eps = betas, eps
        self.state = {}
        self.step_count = 0

    def zero_grad(self, set_to_none: bool = True):
        for g in self.param_groups:
            p = g["params"][0]
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    FUSED_ORDER = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")

    def _state(self, p):
        st = self.state.get(id(p))
        if st is None:
            st = self.state[id(p)] = {"exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)}
        return st

    def fused_backward_args(self, tensors: Dict[str, torch.Tensor]) -> "L.GsrFusedAdam":
        """GsrFusedAdam for the optimizer-in-backward mode of gsr_backward: counts as this optimizer's next step.
        `tensors` are the parameter tensors the rasterizer saved, by group name; they must be the optimizer's own."""
        by_name = {g.get("name"): g for g in self.param_groups}
        if set(by_name) != set(self.FUSED_ORDER):
            raise RuntimeError(f"fused_adam: optimizer groups must be named {self.FUSED_ORDER}, got {tuple(by_name)}")
        fa = L.GsrFusedAdam()
        fa.beta1, fa.beta2, fa.eps = float(self.betas[0]), float(self.betas[1]), float(self.eps)
        for k, name in enumerate(self.FUSED_ORDER):
            g = by_name[name]
            p = g["params"][0]
            t = tensors[name]
            if t.data_ptr() != p.data_ptr() or t.numel() != p.numel() or p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError(f"fused_adam: group '{name}' is not the contiguous float32 tensor that was rasterized")
            st = self._state(p)
            fa.lr[k] = float(g["lr"])
            fa.exp_avg[k], fa.exp_avg_sq[k] = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
        self.step_count += 1
        fa.step = self.step_count
        return fa

    @torch.no_grad()
    def step(self):
        lib = L.load()
        live = [g for g in self.param_groups if g["params"][0].grad is not None]
        if not live:
            return
        self.step_count += 1
        arr = (L.GsrAdamTensor * len(live))()
        keep = []
        dev = live[0]["params"][0].device
        if dev.type != "cuda":
            raise RuntimeError("FusedAdam: parameters must be on a ROCm/HIP device (no CPU fallback)")
        for k, g in enumerate(live):
            p = g["params"][0]
            assert p.is_contiguous() and p.dtype == torch.float32
            st = self._state(p)
            grad = p.grad.contiguous()
            keep.append(grad)
            arr[k].param, arr[k].grad = p.data_ptr(), grad.data_ptr()
            arr[k].exp_avg, arr[k].exp_avg_sq = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            arr[k].n, arr[k].lr = p.numel(), float(g["lr"])
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            L.check(lib.gsr_adam_step(arr, len(live), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                      self.step_count, C.c_void_p(stream)), "gsr_adam_step")
