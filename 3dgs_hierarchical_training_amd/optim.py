"""Multi-tensor Adam on top of gsr_adam_step (one HIP launch for all parameter groups).

Stands where the reference builds `torch.optim.Adam(l, lr=0.0, eps=1e-15)` with one group per parameter tensor
and per-group learning rates (/root/reference/scene/gaussian_model_ht.py:275-289), and keeps the object protocol the
reference's model code relies on, so that its densification / pruning / opacity-reset surgery works on it
unchanged (densification-aware state):

  * `param_groups`: list of dicts with "params" (one tensor), "lr", "name" -- `update_learning_rate` rewrites
    `group["lr"]` every iteration (gaussian_model_ht.py:388-395); the value is read at each step;
  * `state`: dict keyed by the parameter tensor, entries {"step", "exp_avg", "exp_avg_sq"} -- exactly what
    `replace_tensor_to_optimizer` / `_prune_optimizer` / `cat_tensors_to_optimizer` (:532-607) get, slice,
    concatenate, delete and re-insert under a new nn.Parameter;
  * `step()` / `zero_grad(set_to_none=True)` as called at /root/reference/trainer/ht3dgs_trainer.py:159-166;
  * `state_dict()` / `load_state_dict()` in torch.optim's layout (capture / restore, gaussian_model_ht.py:102,124),
    interchangeable with a torch.optim.Adam checkpoint.

Same update rule as torch (no amsgrad, no weight decay); the bias corrections use each tensor's own step count.
No CPU path: `step()` needs the parameters on a ROCm/HIP device.
"""
from typing import Dict, List

import torch

from . import _ext as E
from . import _lib as L


def c_len(opt) -> int:
    return getattr(opt, "_plan_groups", -1)


def _step_int(v) -> int:
    return int(v.item()) if torch.is_tensor(v) else int(v)


class FusedAdam:
    FUSED_ORDER = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")

    def __init__(self, param_groups: List[Dict], lr: float = 0.0, betas=(0.9, 0.999), eps: float = 1e-15):
        self.defaults = {"lr": lr, "betas": tuple(betas), "eps": eps}
        self.param_groups = []
        for g in param_groups:
            g = dict(g)
            g["params"] = list(g["params"])
            g.setdefault("lr", lr)
            assert len(g["params"]) == 1, "one tensor per group, as the reference builds them"
            self.param_groups.append(g)
        self.betas, self.eps = tuple(betas), eps
        self.state = {}
        # Optimizer-in-backward bookkeeping.  A render only PLANS the step (fused_step_plan mutates nothing); the backward that
        # applies it -- the C++ autograd node, or the ctypes route -- increments `_commit`, and the step counts of the six groups
        # catch up from it the next time anybody looks (`_reconcile`: plan, step(), state_dict(), step_count).  A forward that
        # never reaches its backward (no_grad render, an exception in the loss, a dropped graph) therefore leaves no trace.
        self._commit = torch.zeros(1, dtype=torch.int64)      # CPU; shared with csrc/torch_ext.cpp by reference
        self._commit_np = self._commit.numpy()
        self._commit_seen = 0                                  # commits already folded into state[...]["step"]
        self._commit_at_step = 0                               # value of the counter at the last step() call
        # Deferred application (round 4; what gsr_autopatch's render uses under the UNMODIFIED trainer): the backward kernel writes
        # the Adam-updated rows and moments into shadow buffers and `step()` ADOPTS them by swapping storages -- no second pass over
        # 1.65 kB per Gaussian -- while everything the trainer may do between backward() and step() keeps the reference's meaning:
        # densify / prune / reset surgery sees the un-updated state and drops the update, a skipped step() leaves the model alone.
        self._pending = None
        self._shadow = {}                                      # group name -> [param, exp_avg, exp_avg_sq] shadow tensors
        self._def_commit = torch.zeros(1, dtype=torch.int64)   # CPU; incremented by the backward that filled the shadows
        self._def_commit_np = self._def_commit.numpy()

    # ---- torch.optim protocol ------------------------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = True):
        self._pending = None          # a deferred update nobody stepped is dropped with the gradients it stands for
        for g in self.param_groups:
            p = g["params"][0]
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _state(self, p):
        st = self.state.get(p)
        if st is None:
            st = self.state[p] = {}
        # entries may have been rebuilt by the caller's surgery; anything missing starts from zero like torch does
        if "step" not in st:
            st["step"] = 0
        if "exp_avg" not in st:
            st["exp_avg"] = torch.zeros_like(p)
        if "exp_avg_sq" not in st:
            st["exp_avg_sq"] = torch.zeros_like(p)
        for k in ("exp_avg", "exp_avg_sq"):
            m = st[k]
            if m.shape != p.shape or m.device != p.device or m.dtype != torch.float32:
                raise RuntimeError(f"FusedAdam: state '{k}' {tuple(m.shape)} does not match its parameter {tuple(p.shape)}")
            if not m.is_contiguous():      # e.g. a boolean-mask slice is contiguous, a strided view is not
                st[k] = m.contiguous()
        return st

    def _reconcile(self):
        """Fold the in-kernel steps applied since the last look into the six groups' step counts."""
        n = int(self._commit_np[0]) - self._commit_seen
        if n <= 0:
            return
        self._commit_seen += n
        for g in self.param_groups:
            if g.get("name") in self.FUSED_ORDER:
                st = self.state.get(g["params"][0])
                if st is not None:
                    st["step"] = _step_int(st.get("step", 0)) + n

    @property
    def _stepped_in_backward(self) -> bool:
        """True when a backward applied this optimizer's step in-kernel since the last step() call."""
        return int(self._commit_np[0]) > self._commit_at_step

    @property
    def step_count(self) -> int:
        """Largest per-tensor step count (all groups advance together in the reference's loop)."""
        self._reconcile()
        return max([_step_int(st.get("step", 0)) for st in self.state.values()], default=0)

    def state_dict(self) -> Dict:
        self._reconcile()
        index, packed = {}, []
        for g in self.param_groups:
            ids = []
            for p in g["params"]:
                index.setdefault(id(p), len(index))
                ids.append(index[id(p)])
            packed.append({**{k: v for k, v in g.items() if k != "params"}, "betas": self.betas, "eps": self.eps,
                           "params": ids})
        st = {}
        for g in self.param_groups:
            for p in g["params"]:
                if p in self.state:
                    s = self.state[p]
                    st[index[id(p)]] = {k: (torch.tensor(float(_step_int(v))) if k == "step" else v) for k, v in s.items()}
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
                if k not in ("params", "betas", "eps"):
                    g[k] = v
            params += list(zip(saved["params"], g["params"]))
        self._reconcile()
        self.state = {}
        for idx, p in params:
            s = sd["state"].get(idx)
            if s is None:
                continue
            self.state[p] = {"step": _step_int(s.get("step", 0)),
                             "exp_avg": s["exp_avg"].to(device=p.device, dtype=torch.float32).clone(),
                             "exp_avg_sq": s["exp_avg_sq"].to(device=p.device, dtype=torch.float32).clone()}

    # ---- the two ways a step is taken ----------------------------------------------------------------------
    # ---- the f_rest group while its moments are zero ---------------------------------------------------------
    # A model is created at SH degree 0 (gaussian_model_ht.py:68) and stays there for 1 000 iterations (:193-195; all of stage A):
    # the 45 f_rest floats per Gaussian then receive an identically zero gradient, their moments are zero, and Adam's update of
    # them is the identity, bit for bit (0 / (0 + eps) = 0) -- yet it is three quarters of the update's traffic.  A plan for a
    # degree-0 render that prepares no degree-1 view therefore hands the kernel NO moment buffers for that group (GsrFusedAdam:
    # the group is skipped), provided the moments are known to be zero: checked once per pair of state tensors (one
    # count_nonzero, a device sync), re-checked when the tensors are replaced (surgery, load_state_dict) or changed in place
    # through torch (their version counters), and given up for good when a render at degree >= 1 is planned.
    def _rest_moments_zero(self, m: torch.Tensor, v: torch.Tensor) -> bool:
        c = getattr(self, "_rest_zero", None)
        if c is not None and c[0] is m and c[1] is v and c[2] == m._version and c[3] == v._version:
            return c[4]
        z = int(torch.count_nonzero(m)) == 0 and int(torch.count_nonzero(v)) == 0
        self._rest_zero = (m, v, m._version, v._version, z)
        self._rest_zero_checks = getattr(self, "_rest_zero_checks", 0) + 1
        return z

    def _rest_touched(self, m: torch.Tensor, v: torch.Tensor):
        self._rest_zero = (m, v, m._version, v._version, False)

    def _plan_moments(self, ms, vs, sh_degree, next_sh_degree):
        if sh_degree is None or int(sh_degree) > 0:
            self._rest_touched(ms[2], vs[2])          # the group's gradient is (or may be) non-zero from here on
            return ms, vs
        if next_sh_degree is not None and int(next_sh_degree) > 0:
            return ms, vs                              # the prepared degree-1 view reads the rows through the update's tile
        if not self._rest_moments_zero(ms[2], vs[2]):
            return ms, vs
        c = getattr(self, "_skip_lists", None)       # (the two lists are rebuilt only when the plan's tensors change)
        if c is None or c[0] is not ms or c[1] is not vs:
            e = torch.empty(0, dtype=torch.float32, device=ms[2].device)
            c = self._skip_lists = (ms, vs, ms[:2] + [e] + ms[3:], vs[:2] + [e] + vs[3:])
        return c[2], c[3]

    @staticmethod
    def _step_lags(steps):
        """(base, lags): the six groups' step counts as the largest one and how far each group is behind it (GsrFusedAdam::step_lag).
        The groups leave lockstep the way they do in the reference: an opacity reset replaces the opacity tensor between backward()
        and optimizer.step(), which then skips it (gaussian_model_ht.py:468-474, ht3dgs_trainer.py:153-160).  The lags ride behind
        the six learning rates (whole numbers); none when all are zero."""
        base = max(steps)
        lags = [float(base - t) for t in steps]
        return int(base), (lags if any(lags) else [])

    def fused_step_plan(self, tensors: Dict[str, torch.Tensor], sh_degree=None, next_sh_degree=None):
        """(exp_avg[6], exp_avg_sq[6], lr[6] (+ step lags[6]), beta1, beta2, eps, step_base, commit) for the optimizer-in-backward mode of
        gsr_backward.  Mutates nothing: `step_base` is the six groups' current step count and `commit` the CPU counter that the
        backward increments when it has applied the update; the 1-based step of that update is step_base + (commits since this
        plan) + 1.  `tensors` are the parameter tensors being rasterized, by group name; they must be the optimizer's own.
        The learning rates are read here, at render time (the reference sets them before the render, ht3dgs_trainer.py:98)."""
        self._reconcile()
        # fast path: nothing was rebuilt since the previous plan (same group dicts, parameter / moment tensors, state entries):
        # the checks below were all made then -- only the step count and the learning rates are taken anew
        c = getattr(self, "_plan_cache", None)
        if c is not None:
            groups, ps, ts_, sts, ms, vs, idx = c
            ok = len(self.param_groups) == c_len(self)
            for k in range(6 if ok else 0):
                g, st = groups[k], sts[k]
                if self.param_groups[idx[k]] is not g or g["params"][0] is not ps[k] or tensors[self.FUSED_ORDER[k]] is not ts_[k] or self.state.get(ps[k]) is not st \
                        or st.get("exp_avg") is not ms[k] or st.get("exp_avg_sq") is not vs[k]:
                    ok = False
                    break
            if ok:
                pm, pv = self._plan_moments(ms, vs, sh_degree, next_sh_degree)
                base, lags = self._step_lags([_step_int(st["step"]) for st in sts])
                return (pm, pv, [float(g["lr"]) for g in groups] + lags, float(self.betas[0]), float(self.betas[1]), float(self.eps),
                        base, self._commit)
            self._plan_cache = None
        by_name = {g.get("name"): g for g in self.param_groups}
        if set(by_name) != set(self.FUSED_ORDER):
            raise RuntimeError(f"fused_adam: optimizer groups must be named {self.FUSED_ORDER}, got {tuple(by_name)}")
        states, steps, lrs = [], [], []
        for name in self.FUSED_ORDER:
            g = by_name[name]
            p = g["params"][0]
            t = tensors[name]
            if t.data_ptr() != p.data_ptr() or t.numel() != p.numel() or p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError(f"fused_adam: group '{name}' is not the contiguous float32 tensor that was rasterized")
            st = self._state(p)
            states.append(st)
            steps.append(_step_int(st["step"]))
            lrs.append(float(g["lr"]))
        for st, t in zip(states, steps):
            st["step"] = int(t)
        step, lags = self._step_lags(steps)
        ms, vs = [st["exp_avg"] for st in states], [st["exp_avg_sq"] for st in states]
        groups = [by_name[name] for name in self.FUSED_ORDER]
        self._plan_cache = (groups, [g["params"][0] for g in groups], [tensors[name] for name in self.FUSED_ORDER], states, ms, vs,
                            [next(i for i, x in enumerate(self.param_groups) if x is g) for g in groups])
        self._plan_groups = len(self.param_groups)
        pm, pv = self._plan_moments(ms, vs, sh_degree, next_sh_degree)
        return (pm, pv, lrs + lags, float(self.betas[0]), float(self.betas[1]), float(self.eps), int(step), self._commit)

    # ---- deferred application ---------------------------------------------------------------------------------------
    def deferred_ready(self, tensors: Dict[str, torch.Tensor]) -> bool:
        """May this render's backward fill the shadows?  The six tensors are this optimizer's own contiguous float32 leaves, all
        want a gradient, and none carries a stale .grad (a trainer that lets gradients accumulate gets torch's accumulation)."""
        by_name = {g.get("name"): g for g in self.param_groups}
        if set(by_name) != set(self.FUSED_ORDER) or len(self.param_groups) != 6:
            return False
        for name in self.FUSED_ORDER:
            p, t = by_name[name]["params"][0], tensors.get(name)
            if t is not p or not p.requires_grad or p.grad is not None or not p.is_contiguous() or p.dtype != torch.float32 or not p.is_cuda:
                return False
        # an earlier render of this model still holds a plan whose backward has not run (loss = f(render A) + f(render B)):
        # that plan stays, THIS render takes the plain gradient route, and step() -- which then finds a .grad beside the committed
        # shadows -- recovers render A's gradient and adds it (`_adopt_pending`, other_grad).  Torch's sum, not the last node's update.
        if self._pending is not None:
            return False
        return True

    def flush_pending_as_grads(self):
        """A deferred update that was neither adopted (`step()`) nor dropped (`zero_grad()`) when the next render arrives: the trainer
        is letting gradients accumulate.  Give it what torch would hold -- the gradient, recovered from the shadow first moment
        (m' = b1 m + (1 - b1) g) -- and step aside: `deferred_ready` then sees a .grad and the render takes the plain route."""
        pend = self._pending
        if pend is None:
            return
        if int(self._def_commit_np[0]) <= pend["commit_at_plan"]:
            # not committed: its backward has not run -- yet.  It may still (two renders feeding one backward), so the plan is KEPT:
            # `deferred_ready` sends the arriving render down the plain route; a graph that was simply dropped costs the same one
            # plain iteration, and step() / zero_grad() clear the plan either way
            return
        self._pending = None
        b1 = float(self.betas[0])
        with torch.no_grad():
            for k, name in enumerate(self.FUSED_ORDER):
                p = pend["params"][k]
                if self.state.get(p) is None or self.state[p].get("exp_avg") is not pend["m"][k]:
                    continue
                # (a group the kernel skipped -- f_rest at degree 0 with zero moments -- had an identically zero gradient)
                g = torch.zeros_like(p) if pend["skipped"][k] else (pend["mo"][k] - b1 * pend["m"][k]) / (1.0 - b1)
                p.grad = g if p.grad is None else p.grad + g

    def deferred_step_plan(self, tensors: Dict[str, torch.Tensor], sh_degree=None):
        """Like fused_step_plan, for a backward that writes the update into shadow buffers: returns
        (exp_avg[6] + exp_avg_out[6] + param_out[6], exp_avg_sq[6] + exp_avg_sq_out[6], lr (+ lags), b1, b2, eps, step_base, commit)."""
        pm, pv, lr, b1, b2, eps, step, _ = self.fused_step_plan(tensors, sh_degree, None)
        by_name = {g.get("name"): g for g in self.param_groups}
        ps = [by_name[n]["params"][0] for n in self.FUSED_ORDER]
        po, mo, vo, skipped = [], [], [], []
        for k, name in enumerate(self.FUSED_ORDER):
            p = ps[k]
            sh = self._shadow.get(name)
            if sh is None or sh[0].shape != p.shape or sh[0].device != p.device:
                sh = self._shadow[name] = [torch.empty_like(p.detach()), torch.empty_like(p.detach()), torch.empty_like(p.detach())]
            skip = pm[k].numel() == 0            # (the f_rest group while its moments are zero: left alone, nothing to adopt)
            skipped.append(skip)
            po.append(sh[0]); mo.append(sh[1]); vo.append(sh[2])
        st = [self.state[p] for p in ps]
        self._pending = {"params": ps, "m": [s_["exp_avg"] for s_ in st], "v": [s_["exp_avg_sq"] for s_ in st], "po": po, "mo": mo, "vo": vo,
                         "skipped": skipped, "commit_at_plan": int(self._def_commit_np[0]),
                         "lr": [float(by_name[n]["lr"]) for n in self.FUSED_ORDER]}
        return (list(pm) + mo + po, list(pv) + vo, lr, b1, b2, eps, step, self._def_commit)

    def _adopt_pending(self) -> bool:
        """step() of a deferred update: swap the shadows in.  False when there is nothing valid to adopt."""
        pend, self._pending = self._pending, None
        if pend is None or int(self._def_commit_np[0]) <= pend["commit_at_plan"]:
            return False                      # no backward filled the shadows (no_grad render, dropped graph)
        by_name = {g.get("name"): g for g in self.param_groups}
        if set(by_name) != set(self.FUSED_ORDER):
            return False
        other_grad = any(by_name[n]["params"][0] is pend["params"][k] and by_name[n]["params"][0].grad is not None
                         for k, n in enumerate(self.FUSED_ORDER))
        if other_grad or [float(by_name[n]["lr"]) for n in self.FUSED_ORDER] != pend["lr"]:
            # a learning rate changed between the render and step() (the reference sets them before the render), or another backward
            # left a .grad on these parameters: the shadows do not hold the step torch would take -- recover the gradients (they
            # accumulate onto whatever .grad is there) and let the plain step below do it
            self._pending = pend
            self.flush_pending_as_grads()
            return False
        # per group, as torch steps per parameter: a group whose tensor (or moments) the trainer replaced between backward() and
        # step() has no gradient in the reference and is not stepped (opacity reset: that one group; densification: all six);
        # the others adopt their shadows
        adopted = False
        with torch.no_grad():
            for k, name in enumerate(self.FUSED_ORDER):
                p = by_name[name]["params"][0]
                st = self.state.get(p)
                if p is not pend["params"][k] or p.grad is not None or st is None:
                    continue
                if pend["skipped"][k]:            # (f_rest while its moments are zero: the update was the identity; count the step)
                    st["step"] = _step_int(st.get("step", 0)) + 1
                    continue
                if st.get("exp_avg") is not pend["m"][k] or st.get("exp_avg_sq") is not pend["v"][k]:
                    continue
                old = p.data
                p.data = pend["po"][k]
                sh = self._shadow[name]
                sh[0], sh[1], sh[2] = old, st["exp_avg"], st["exp_avg_sq"]
                st["exp_avg"], st["exp_avg_sq"] = pend["mo"][k], pend["vo"][k]
                st["step"] = _step_int(st["step"]) + 1
                adopted = True
                c = self._plan_cache       # keep the next plan on its fast path: the cached moment lists follow the swap
                if c is not None and c[3][k] is st:
                    c[4][k], c[5][k] = st["exp_avg"], st["exp_avg_sq"]
        self._skip_lists = None            # (built from the moment lists)
        return adopted

    def fused_backward_args(self, tensors: Dict[str, torch.Tensor], sh_degree=None, next_sh_degree=None) -> "L.GsrFusedAdam":
        """The plan as a GsrFusedAdam struct for the update that is applied NOW (ctypes binding: called from its backward, which
        calls `fused_backward_applied()` once gsr_backward has returned)."""
        m, v, lrs, b1, b2, eps, step, _ = self.fused_step_plan(tensors, sh_degree, next_sh_degree)
        fa = L.GsrFusedAdam()
        fa.beta1, fa.beta2, fa.eps, fa.step = b1, b2, eps, step + 1
        for k in range(6):
            fa.lr[k] = lrs[k]
            fa.step_lag[k] = int(lrs[6 + k]) if len(lrs) == 12 else 0
            fa.exp_avg[k] = m[k].data_ptr() if m[k].numel() else None       # (no buffers: the group is skipped)
            fa.exp_avg_sq[k] = v[k].data_ptr() if v[k].numel() else None
        return fa

    def fused_backward_applied(self):
        self._commit_np[0] += 1

    @torch.no_grad()
    def step(self):
        self._reconcile()
        if self._pending is not None and self._adopt_pending():
            return
        live = [g for g in self.param_groups if g["params"][0].grad is not None]
        stepped = self._stepped_in_backward
        self._commit_at_step = int(self._commit_np[0])
        if not live:
            return
        for g in live:
            if g.get("name") == "f_rest" and g["params"][0] in self.state:
                st = self.state[g["params"][0]]
                if "exp_avg" in st and "exp_avg_sq" in st:
                    self._rest_touched(st["exp_avg"], st["exp_avg_sq"])   # a .grad reached the group: its moments may leave zero
        if stepped and any(g.get("name") in self.FUSED_ORDER for g in live):
            # the in-kernel step of this iteration has already been applied; a .grad on the same parameters (a second loss
            # term, a second backward) would be a second Adam update with a second step count
            raise RuntimeError("FusedAdam.step(): this iteration's step was already applied inside backward() (fused_adam), but "
                               "the parameters also carry a .grad -- render without fused_adam when other loss terms reach them")
        dev = live[0]["params"][0].device
        if dev.type != "cuda":
            raise RuntimeError("FusedAdam: parameters must be on a ROCm/HIP device (no CPU fallback)")
        ops = E.load()
        by_step = {}
        for g in live:
            p = g["params"][0]
            if not (p.is_contiguous() and p.dtype == torch.float32):
                raise RuntimeError("FusedAdam: parameters must be contiguous float32")
            st = self._state(p)
            st["step"] = _step_int(st["step"]) + 1
            by_step.setdefault(st["step"], []).append((g, p, st))
        for step, items in by_step.items():      # one launch; several only if tensors joined the optimizer at different times
            ops.adam_step([p for _, p, _ in items], [p.grad for _, p, _ in items], [st["exp_avg"] for _, _, st in items],
                          [st["exp_avg_sq"] for _, _, st in items], [float(g["lr"]) for g, _, _ in items], float(self.betas[0]),
                          float(self.betas[1]), float(self.eps), int(step))
