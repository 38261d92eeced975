"""CPU: FusedAdam keeps torch.optim's object protocol (no step is taken here -- stepping needs the GPU).

The reference's model code edits optimizer state directly (/root/reference/scene/gaussian_model_ht.py:532-607:
`optimizer.state.get(group['params'][0])`, slicing / concatenating exp_avg and exp_avg_sq, `del state[old]`,
`state[new] = stored_state`).  These tests run that call pattern against FusedAdam and torch.optim.Adam side
by side, and check that checkpoints are interchangeable in both directions."""
import importlib

import torch
from torch import nn

optim = importlib.import_module("3dgs_hierarchical_training_amd.optim")
ts = importlib.import_module("3dgs_hierarchical_training_amd.train_step")
syn = importlib.import_module("3dgs_hierarchical_training_amd.synthetic")

NAMES = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
SHAPES = {"xyz": (3,), "f_dc": (1, 3), "f_rest": (15, 3), "opacity": (1,), "scaling": (3,), "rotation": (4,)}


def _groups(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [{"params": [nn.Parameter(torch.randn((n,) + SHAPES[k], generator=g))], "lr": 0.001 * (i + 1), "name": k}
            for i, k in enumerate(NAMES)]


def _seed_state(opt, step=7):
    for i, g in enumerate(opt.param_groups):
        p = g["params"][0]
        opt.state[p] = {"step": torch.tensor(float(step)) if isinstance(opt, torch.optim.Adam) else step,
                        "exp_avg": torch.full_like(p, 0.1 * (i + 1)), "exp_avg_sq": torch.full_like(p, 0.01 * (i + 1))}


def _prune_like_the_model_does(opt, keep):
    out = {}
    for group in opt.param_groups:
        stored = opt.state.get(group["params"][0], None)
        if stored is not None:
            stored["exp_avg"] = stored["exp_avg"][keep]
            stored["exp_avg_sq"] = stored["exp_avg_sq"][keep]
            del opt.state[group["params"][0]]
            group["params"][0] = nn.Parameter(group["params"][0][keep].requires_grad_(True))
            opt.state[group["params"][0]] = stored
        else:
            group["params"][0] = nn.Parameter(group["params"][0][keep].requires_grad_(True))
        out[group["name"]] = group["params"][0]
    return out


def _cat_like_the_model_does(opt, ext):
    for group in opt.param_groups:
        e = ext[group["name"]]
        stored = opt.state.get(group["params"][0], None)
        if stored is not None:
            stored["exp_avg"] = torch.cat((stored["exp_avg"], torch.zeros_like(e)), dim=0)
            stored["exp_avg_sq"] = torch.cat((stored["exp_avg_sq"], torch.zeros_like(e)), dim=0)
            del opt.state[group["params"][0]]
            group["params"][0] = nn.Parameter(torch.cat((group["params"][0], e), dim=0).requires_grad_(True))
            opt.state[group["params"][0]] = stored
        else:
            group["params"][0] = nn.Parameter(torch.cat((group["params"][0], e), dim=0).requires_grad_(True))


def test_state_surgery_matches_torch_adam():
    a = optim.FusedAdam(_groups(50), lr=0.0, eps=1e-15)
    b = torch.optim.Adam(_groups(50), lr=0.0, eps=1e-15)
    _seed_state(a)
    _seed_state(b)
    keep = torch.arange(50) % 3 != 0
    ext = {k: torch.ones((5,) + SHAPES[k]) for k in NAMES}
    for opt in (a, b):
        _prune_like_the_model_does(opt, keep)
        _cat_like_the_model_does(opt, ext)
    assert len(a.state) == len(b.state) == 6
    for ga, gb in zip(a.param_groups, b.param_groups):
        pa, pb = ga["params"][0], gb["params"][0]
        assert pa.shape == pb.shape and pa.shape[0] == int(keep.sum()) + 5 and torch.equal(pa, pb)
        sa, sb = a._state(pa), b.state[pb]
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
        assert sa["exp_avg"].is_contiguous() and int(sa["step"]) == int(sb["step"]) == 7
    assert a.step_count == 7


def test_learning_rate_is_read_from_the_group_each_step():
    a = optim.FusedAdam(_groups(4), lr=0.0, eps=1e-15)
    for g in a.param_groups:           # what update_learning_rate does (gaussian_model_ht.py:388-395)
        if g["name"] == "xyz":
            g["lr"] = 1.25e-4
    assert [g["lr"] for g in a.param_groups if g["name"] == "xyz"] == [1.25e-4]
    assert a.param_groups[0]["params"][0].grad is None
    a.step()                            # nothing has a gradient: no launch, no library needed
    a.zero_grad(set_to_none=True)
    assert a.step_count == 0


def test_checkpoints_are_interchangeable_with_torch_adam():
    a = optim.FusedAdam(_groups(20, seed=1), lr=0.0, eps=1e-15)
    b = torch.optim.Adam(_groups(20, seed=1), lr=0.0, eps=1e-15)
    _seed_state(b, step=11)
    a.load_state_dict(b.state_dict())           # torch checkpoint -> FusedAdam
    for ga, gb in zip(a.param_groups, b.param_groups):
        sa, sb = a.state[ga["params"][0]], b.state[gb["params"][0]]
        assert sa["step"] == 11 and torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
        assert ga["lr"] == gb["lr"] and ga["name"] == gb["name"]
    c = torch.optim.Adam(_groups(20, seed=1), lr=0.0, eps=1e-15)
    c.load_state_dict(a.state_dict())           # FusedAdam checkpoint -> torch
    for gc, gb in zip(c.param_groups, b.param_groups):
        sc, sb = c.state[gc["params"][0]], b.state[gb["params"][0]]
        assert float(sc["step"]) == 11.0 and torch.equal(sc["exp_avg"], sb["exp_avg"])
        assert gc["lr"] == gb["lr"]


def test_gaussian_params_surgery_on_cpu():
    """prune_points / densification_postfix / reset_opacity of the benchmark's parameter store (CPU tensors are fine
    for the bookkeeping; only step() needs the device)."""
    sc = syn.make_scene(64, 32, 32, sh_degree=3, seed=3)
    for kind in ("hip", "torch"):
        p = ts.GaussianParams(sc, torch.device("cpu"), optimizer=kind)
        for g in p.optimizer.param_groups:
            q = g["params"][0]
            p.optimizer.state[q] = {"step": torch.tensor(3.0) if kind == "torch" else 3, "exp_avg": torch.ones_like(q),
                                    "exp_avg_sq": torch.ones_like(q)}
        mask = torch.zeros(64, dtype=torch.bool)
        mask[::4] = True
        p.prune_points(mask)
        assert p.num_points == 48 and p._features_rest.shape == (48, 15, 3) and p._xyz.requires_grad
        new = {"xyz": torch.zeros(8, 3), "f_dc": torch.zeros(8, 1, 3), "f_rest": torch.zeros(8, 15, 3),
               "opacity": torch.zeros(8, 1), "scaling": torch.zeros(8, 3), "rotation": torch.zeros(8, 4)}
        p.densification_postfix(new)
        assert p.num_points == 56
        st = p.optimizer.state[p._xyz]
        assert st["exp_avg"].shape == (56, 3) and float(st["exp_avg"][:48].min()) == 1.0 and float(st["exp_avg"][48:].abs().max()) == 0.0
        p.reset_opacity()
        assert float(p.get_opacity.detach().max()) <= 0.0100001
        so = p.optimizer.state[p._opacity]
        assert float(so["exp_avg"].abs().max()) == 0.0 and int(so["step"]) == 3
        assert set(id(g["params"][0]) for g in p.optimizer.param_groups) == {id(getattr(p, a)) for a in p._GROUP_ATTR.values()}


def test_plan_leaves_f_rest_out_only_while_its_moments_are_known_to_be_zero():
    """The planning side of GsrFusedAdam's "no moment buffers for f_rest" (host logic only: the kernels are not involved).
    Degree 0 and no degree-1 view prepared: the plan carries EMPTY tensors for the group, after ONE validation of the moments;
    surgery that keeps them zero (prune, cat of zeros: the model file's own calls) is re-validated and keeps the skip; an
    in-place change through torch, a plan at degree >= 1 or an unknown degree end it."""
    opt = optim.FusedAdam(_groups(40), lr=0.0, eps=1e-15)
    tensors = lambda: {g["name"]: g["params"][0] for g in opt.param_groups}
    ms, vs = opt.fused_step_plan(tensors(), 0, 0)[:2]
    assert ms[2].numel() == 0 and vs[2].numel() == 0 and all(ms[k].numel() > 0 for k in (0, 1, 3, 4, 5))
    assert opt.fused_step_plan(tensors(), 0, None)[0][2].numel() == 0 and opt._rest_zero_checks == 1
    assert opt.fused_step_plan(tensors(), 0, 1)[0][2].numel() > 0                 # the degree-1 hand-over reads the rows
    assert opt.fused_step_plan(tensors(), 0, 0)[0][2].numel() == 0 and opt._rest_zero_checks == 1
    keep = torch.arange(40) % 3 != 0
    _prune_like_the_model_does(opt, keep)
    assert opt.fused_step_plan(tensors(), 0, 0)[0][2].numel() == 0 and opt._rest_zero_checks == 2   # new tensors: validated again
    _cat_like_the_model_does(opt, {k: torch.randn((5,) + SHAPES[k]) for k in NAMES})
    assert opt.fused_step_plan(tensors(), 0, 0)[0][2].numel() == 0 and opt._rest_zero_checks == 3
    rest = next(g["params"][0] for g in opt.param_groups if g["name"] == "f_rest")
    opt.state[rest]["exp_avg_sq"][3, 2, 1] = 1e-12                                 # in place through torch: version counter
    assert opt.fused_step_plan(tensors(), 0, 0)[0][2].numel() > 0 and opt._rest_zero_checks == 4
    opt.state[rest]["exp_avg_sq"].zero_()
    assert opt.fused_step_plan(tensors(), 0, 0)[0][2].numel() == 0                 # ... and back
    assert opt.fused_step_plan(tensors(), 1, 1)[0][2].numel() > 0                  # a degree-1 step: gradients will reach the group
    assert opt.fused_step_plan(tensors(), 0, 0)[0][2].numel() > 0                  # for good (no re-validation: the mark is "touched")
    opt2 = optim.FusedAdam(_groups(8), lr=0.0, eps=1e-15)
    t2 = {g["name"]: g["params"][0] for g in opt2.param_groups}
    assert opt2.fused_step_plan(t2)[0][2].numel() > 0 and opt2.fused_step_plan(t2, 0, 0)[0][2].numel() > 0   # unknown degree = touched
