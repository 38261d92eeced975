"""gsr_autopatch (opt-in fast path for the unmodified reference trainer): the dispatch and import-hook logic, on CPU."""
import importlib
import os
import sys

import pytest
import torch

NAMES = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]


@pytest.fixture
def autopatch():
    import gsr_autopatch
    gsr_autopatch.apply()
    yield gsr_autopatch
    gsr_autopatch.remove()


def test_only_the_reference_optimizer_is_redirected(autopatch):
    stock = autopatch._ORIG_ADAM
    # CPU parameters: the six-group construction of gaussian_model_ht.py:275-289 stays stock Adam (FusedAdam is GPU-only)
    ps = [torch.nn.Parameter(torch.zeros(4, 3)) for _ in NAMES]
    o = torch.optim.Adam([{"params": [p], "lr": 0.1, "name": n} for p, n in zip(ps, NAMES)], lr=0.0, eps=1e-15)
    assert type(o) is stock
    # anything that is not that construction is never touched
    assert type(torch.optim.Adam([torch.nn.Parameter(torch.zeros(2))], lr=1e-3)) is stock
    assert type(torch.optim.Adam([{"params": [ps[0]], "name": "xyz"}], lr=1e-3)) is stock
    assert autopatch._wants_fused([{"params": [p], "name": n} for p, n in zip(ps, NAMES)], {}) is False          # CPU tensors
    assert autopatch._wants_fused([{"params": [p], "name": n + "_"} for p, n in zip(ps, NAMES)], {}) is False    # other names


def test_remove_restores_torch(autopatch):
    stock = autopatch._ORIG_ADAM
    assert torch.optim.Adam is not stock
    autopatch.remove()
    assert torch.optim.Adam is stock and autopatch._FINDER not in sys.meta_path
    autopatch.apply()


@pytest.mark.skipif(not os.path.isdir("/root/reference/trainer"), reason="the reference tree only exists in the authoring container")
def test_import_hook_patches_the_reference_loss_module(autopatch):
    """`trainer.losses` imported AFTER gsr_autopatch gets its Loss.forward replaced, with no reference file edited; remove() puts
    the original back."""
    sys.path.insert(0, "/root/reference")
    try:
        sys.modules.pop("trainer.losses", None)
        sys.modules.pop("trainer", None)
        TL = importlib.import_module("trainer.losses")
        assert TL.Loss.forward is autopatch.loss_forward
        autopatch.remove()
        assert TL.Loss.forward is not autopatch.loss_forward and TL.Loss.forward.__qualname__ == "Loss.forward"
        autopatch.apply()
        assert TL.Loss.forward is autopatch.loss_forward      # already imported: patched in place
    finally:
        sys.path.remove("/root/reference")
        sys.modules.pop("trainer.losses", None)
        sys.modules.pop("trainer", None)
