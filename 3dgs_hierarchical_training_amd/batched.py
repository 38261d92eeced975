"""B independent models trained in ONE launch chain (include/gsr.h GsrBatch).

Stage A of the reference -- ~70 % of a scene's render calls -- fits F - 1 single-image models that share nothing
(`for fidx in range(1, seq_len): compute_relative_pose(fidx, fidx - 1)`, /root/reference/trainer/ht3dgs_trainer.py:697-698,
:336-431; the authors note the independence, /root/reference/README.md:131).  One such model (~130 k Gaussians, one 980x545
image) is a chain of ~17 dependent kernels whose sorts, scans and per-Gaussian passes each fill a fraction of an MI355X.  Here B
of them share one parameter store: model b owns the 128-Gaussian blocks [first_block[b], first_block[b + 1]) (each model is
padded to a multiple of 128 with Gaussians that can never be drawn), every kernel of the chain runs once over all of them, and
the B images come back as [B,3,H,W].  Each model's image, radii and parameter updates are bit-identical with training it alone
(tests/test_gpu_batched.py): the B images form one tall tile grid, so a tile's list only ever holds its own model's Gaussians.
"""
from typing import Dict, List, Sequence

import torch

from .rasterizer import GaussianRasterizationSettings
from .train_step import GaussianParams

BLOCK = 128


def _pad_rows(t: torch.Tensor, n_pad: int, fill: torch.Tensor) -> torch.Tensor:
    if n_pad == 0:
        return t
    return torch.cat((t, fill.to(t.dtype).expand((n_pad,) + tuple(t.shape[1:]))), dim=0)


def concat_scenes(scenes: Sequence[Dict]) -> Dict:
    """One scene dict holding the models back to back, each padded to a multiple of 128 Gaussians.  Padding rows sit far behind
    every camera with an opacity below the 1/255 threshold: culled by the near plane and by the exact tile test alike, they get no
    instance, no gradient, and Adam leaves them where they are (zero gradient, zero moments).  Returns the scene with
    `first_block` (B + 1 block offsets) and `counts` (the models' true sizes)."""
    parts, first_block, counts = {k: [] for k in ("means3D", "shs", "scales", "rotations", "opacities")}, [0], []
    for sc in scenes:
        n = sc["means3D"].shape[0]
        n_pad = (-n) % BLOCK
        parts["means3D"].append(_pad_rows(sc["means3D"], n_pad, torch.tensor([[0.0, 0.0, -1.0e3]])))
        parts["shs"].append(_pad_rows(sc["shs"], n_pad, torch.zeros(1, sc["shs"].shape[1], 3)))
        parts["scales"].append(_pad_rows(sc["scales"], n_pad, torch.full((1, 3), 1e-3)))
        parts["rotations"].append(_pad_rows(sc["rotations"], n_pad, torch.tensor([[1.0, 0.0, 0.0, 0.0]])))
        parts["opacities"].append(_pad_rows(sc["opacities"], n_pad, torch.full((1, 1), 1e-4)))
        counts.append(n)
        first_block.append(first_block[-1] + (n + n_pad) // BLOCK)
    out = dict(scenes[0])
    for k, v in parts.items():
        out[k] = torch.cat(v, dim=0).contiguous()
    out["first_block"], out["counts"] = first_block, counts
    return out


def batch_settings(views: Sequence[GaussianRasterizationSettings], device) -> GaussianRasterizationSettings:
    """One settings tuple for B renders: the cameras stacked ([B,4,4], [B,4,4], [B,3]); image size, field of view, background,
    degree and scale modifier are shared and must agree."""
    v0 = views[0]
    for v in views[1:]:
        if (v.image_height, v.image_width, v.tanfovx, v.tanfovy, v.sh_degree, v.scale_modifier) != \
                (v0.image_height, v0.image_width, v0.tanfovx, v0.tanfovy, v0.sh_degree, v0.scale_modifier):
            raise ValueError("batch_settings: the views of a batch share image size, field of view, SH degree and scale modifier")
    return v0._replace(viewmatrix=torch.stack([v.viewmatrix.to(device).float() for v in views]).contiguous(),
                       projmatrix=torch.stack([v.projmatrix.to(device).float() for v in views]).contiguous(),
                       campos=torch.stack([v.campos.to(device).float() for v in views]).contiguous())


class BatchedGaussianParams(GaussianParams):
    """`GaussianParams` over B models stored back to back (`first_block`, `counts`); one optimizer steps them all (they advance
    in lockstep, as B stage-A fits started together do).  `train_step.render` / `train_step.train_step` recognise the batch by
    `first_block` and return [B,3,H,W] images; targets are [B,3,H,W]."""

    def __init__(self, scenes: Sequence[Dict], device, spatial_lr_scale: float = 1.0, optimizer: str = "hip"):
        sc = concat_scenes(scenes)
        super().__init__(sc, device, spatial_lr_scale=spatial_lr_scale, optimizer=optimizer)
        self.first_block: List[int] = list(sc["first_block"])
        self.counts: List[int] = list(sc["counts"])

    @property
    def num_models(self) -> int:
        return len(self.counts)

    def model_rows(self, b: int) -> slice:
        """Rows of model b's own (un-padded) Gaussians."""
        lo = self.first_block[b] * BLOCK
        return slice(lo, lo + self.counts[b])

    def model_raw(self, b: int) -> Dict[str, torch.Tensor]:
        """Model b's six raw tensors (views of the store, detached): what `capture()` would hold for it."""
        r = self.model_rows(b)
        return {k: getattr(self, k).detach()[r] for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")}

    # resizing one model of a batch would move every later model's blocks: stage A never densifies (densify=False at
    # /root/reference/trainer/ht3dgs_trainer.py:298)
    def prune_points(self, mask):
        raise RuntimeError("BatchedGaussianParams: the models of a batch cannot be resized")

    def densification_postfix(self, new):
        raise RuntimeError("BatchedGaussianParams: the models of a batch cannot be resized")
