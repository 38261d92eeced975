"""The reference's per-frame pose objects on the kernels (VERDICT r5 "missing #3").

The unmodified trainer keeps one lietorch `LieGroupParameter(SE3(pose7))` per frame (`HTGaussianModel.P`,
/root/reference/scene/gaussian_model_ht.py:346-386), moves the points through it before every render
(`get_xyz`: `P[k].retr().act(xyz)`, :135-148) and steps it with an Adam of its own after every render
(`camera_optimizer[k]`, :296-311, stepped at /root/reference/trainer/ht3dgs_trainer.py:162-166; stage A keeps the one pose in
`gaussians.optimizer`, `training_setup_fix_position(gaussian_rot=False)`, gaussian_model_ht.py:321-333).  A LieGroupParameter IS
a float32 tensor of the group's tangent shape ([1,6], zeros at construction: lietorch/groups.py `LieGroupParameter.__new__`)
with the group element in `.group` (`.group.data` = [1,7]: tx ty tz qx qy qz qw); `retr()` = Exp(tensor) * group.

This module holds the two pieces `gsr_autopatch` puts under those statements, neither of which touches an N-sized tensor or runs
a torch matrix chain:
  * `pose_matrix(p)`: the [3,4] matrix of Exp(p) * p.group as ONE autograd node (`torch.ops.gsr.pose_matrix`: one one-wave
    kernel forward, one backward that leaves dL/d(tangent) in `p.grad`) -- handed to the rasterizer as `points_transform`;
  * `FusedPoseAdam`: what `torch.optim.Adam([{'params': [P[k]], 'lr': ..., 'name': 'R'}], lr=0.0, eps=1e-15)` constructs
    while the patch is applied: torch.optim's object protocol (`param_groups`, `state`, `step`, `zero_grad`, `state_dict`,
    `load_state_dict`), the update itself one `gsr_adam_step` launch over the six numbers (torch's rule, float32).
Anything that is not shaped like a lietorch SE3 parameter is left to the original code (`is_lie_pose` says which).
No CPU path: the kernels need the tensors on a ROCm/HIP device.
"""
from typing import Dict, List

import torch

from . import pose as _pose


def is_lie_pose(p) -> bool:
    """`p` looks like `LieGroupParameter(SE3(pose7[None]))`: a float32 tensor of six numbers that wants a gradient, with an SE3
    group element of seven numbers in `.group.data`.  Duck-typed -- lietorch is not importable where this library is built."""
    if not torch.is_tensor(p) or p.dtype != torch.float32 or p.numel() != 6 or not p.requires_grad:
        return False
    grp = getattr(p, "group", None)
    data = getattr(grp, "data", None)
    return grp is not None and type(grp).__name__ == "SE3" and torch.is_tensor(data) and data.numel() == 7 and data.dtype == torch.float32


def base_matrix(p) -> torch.Tensor:
    """[3,4] matrix of the group element `p.group` (pose7 = t, q_xyzw), on p's device; built once per group element with a
    handful of torch statements on seven numbers and kept on the parameter object until the element is replaced or written."""
    data = p.group.data
    c = getattr(p, "_gsr_base", None)
    if c is not None and c[0] is data and c[1] == data._version and c[2].device == p.device:
        return c[2]
    with torch.no_grad():
        B = _pose.pose7_to_matrix(data.detach().reshape(7).to(device=p.device, dtype=torch.float64))[:3].to(torch.float32).contiguous()
    try:
        p._gsr_base = (data, data._version, B)
    except Exception:
        pass
    return B


def pose_matrix(p, ops) -> torch.Tensor:
    """Exp(p) * p.group as a [3,4] tensor linked to `p` by one autograd node (see the module docstring)."""
    return ops.pose_matrix(p, base_matrix(p))


class FusedPoseAdam:
    """torch.optim.Adam over lietorch pose parameters (one or more groups of one `LieGroupParameter` each; no amsgrad, no
    weight decay), one launch per `step()`."""

    def __init__(self, param_groups: List[Dict], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        self.defaults = {"lr": lr, "betas": tuple(betas), "eps": eps}
        self.param_groups = []
        for g in param_groups:
            g = dict(g)
            g["params"] = list(g["params"])
            g.setdefault("lr", lr)
            g.setdefault("betas", tuple(betas))
            g.setdefault("eps", eps)
            self.param_groups.append(g)
        self.betas, self.eps = tuple(betas), eps
        self.state = {}

    def zero_grad(self, set_to_none: bool = True):
        for g in self.param_groups:
            for p in g["params"]:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()

    def _state(self, p):
        st = self.state.get(p)
        if st is None:
            st = self.state[p] = {}
        if "step" not in st:
            st["step"] = 0
        for k in ("exp_avg", "exp_avg_sq"):
            if k not in st:
                st[k] = torch.zeros(p.shape, dtype=torch.float32, device=p.device)
        return st

    @torch.no_grad()
    def step(self):
        from . import _ext as E
        by_step = {}
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedPoseAdam: parameters must be on a ROCm/HIP device (no CPU fallback)")
                if not (p.is_contiguous() and p.dtype == torch.float32 and p.grad.is_contiguous() and p.grad.dtype == torch.float32):
                    raise RuntimeError("FusedPoseAdam: parameters and gradients must be contiguous float32")
                st = self._state(p)
                st["step"] = int(st["step"].item() if torch.is_tensor(st["step"]) else st["step"]) + 1
                b = tuple(g.get("betas", self.betas))
                by_step.setdefault((st["step"], float(b[0]), float(b[1]), float(g.get("eps", self.eps))), []).append((g, p, st))
        if not by_step:
            return
        ops = E.load()
        for (step, b1, b2, eps), items in by_step.items():     # (one launch: the reference builds one group per optimizer)
            # (a LieGroupParameter is a tensor subclass with __torch_function__ disabled: `.data` is the plain tensor over the same storage)
            ops.adam_step([p.data for _, p, _ in items], [p.grad for _, p, _ in items], [st["exp_avg"] for _, _, st in items],
                          [st["exp_avg_sq"] for _, _, st in items], [float(g["lr"]) for g, _, _ in items], b1, b2, eps, int(step))

    def state_dict(self) -> Dict:
        index, packed = {}, []
        for g in self.param_groups:
            ids = []
            for p in g["params"]:
                index.setdefault(id(p), len(index))
                ids.append(index[id(p)])
            packed.append({**{k: v for k, v in g.items() if k != "params"}, "params": ids})
        st = {}
        for g in self.param_groups:
            for p in g["params"]:
                if p in self.state:
                    s = self.state[p]
                    st[index[id(p)]] = {k: (torch.tensor(float(int(v))) if k == "step" else v) for k, v in s.items()}
        return {"state": st, "param_groups": packed}

    def load_state_dict(self, sd: Dict):
        groups = sd["param_groups"]
        if len(groups) != len(self.param_groups):
            raise ValueError("loaded state dict has a different number of parameter groups")
        params = []
        for g, saved in zip(self.param_groups, groups):
            if len(saved["params"]) != len(g["params"]):
                raise ValueError("loaded state dict contains a parameter group that doesn't match the size of optimizer's group")
            for k, v in saved.items():
                if k != "params":
                    g[k] = v
            params += list(zip(saved["params"], g["params"]))
        self.state = {}
        for idx, p in params:
            s = sd["state"].get(idx)
            if s is None:
                continue
            step = s.get("step", 0)
            self.state[p] = {"step": int(step.item() if torch.is_tensor(step) else step),
                             "exp_avg": s["exp_avg"].to(device=p.device, dtype=torch.float32).clone(),
                             "exp_avg_sq": s["exp_avg_sq"].to(device=p.device, dtype=torch.float32).clone()}
