"""Synthetic frame sequence + the frame partition of the hierarchical trainer (harness data for BASELINE config 4).

There are no datasets in this environment, so a "video" is made from a ground-truth Gaussian cloud seen from F cameras
on a smooth trajectory; the target image of frame f is the ground-truth cloud rendered by this library's own rasterizer
(on the device, once).  What mirrors the reference:
  * `partition`   /root/reference/trainer/ht3dgs_trainer.py:1338-1395, the evenly-sampled strategy (:1379-1395): every
                  level halves each frame list with two frames of overlap; written for any level (the reference
                  unrolls three);
  * leaf coordinates: a leaf model lives in the camera frame of its first frame (`start_fidx`, :733-735), frame f is
                  seen through `get_RT(f)` = the chain of relative poses from the start frame (:738-741);
  * `camera_for`  what `load_viewpoint_cam(fidx, pose=...)` hands the renderer: the co3d-style camera of
                  /root/reference/scene/cameras.py:76-98 for a world-to-camera pose (synthetic.make_camera).
"""
import math
from typing import Dict, List

import torch

from . import synthetic as syn
from .rasterizer import GaussianRasterizationSettings


def partition(n: int, level: int, overlap: int = 2) -> Dict[int, List[List[int]]]:
    """{level -> list of frame lists}; result[0] == [range(n)], result[k] has 2^k lists, neighbours share `overlap`
    frames (for overlap = 2: first half [:m//2+1], second half [m//2-1:], as :1386-1394)."""
    result = {0: [list(range(n))]}
    for lv in range(1, level + 1):
        result[lv] = []
        for ind in result[lv - 1]:
            m = len(ind)
            assert m >= 2 + overlap, f"{n} frames are too few for level {level}"
            result[lv].append(ind[:m // 2 + overlap // 2])
            result[lv].append(ind[m // 2 - (overlap - overlap // 2):])
    return result


class FrameSequence:
    """F frames of one static ground-truth cloud.  `w2c[f]` = world-to-camera of frame f; frame 0 is the identity."""

    def __init__(self, n_frames: int, gt_gaussians: int, W: int, H: int, device, sh_degree: int = 3, seed: int = 0,
                 step_angle: float = 0.006, step_shift: float = 0.02):
        self.F, self.W, self.H, self.device, self.sh_degree = n_frames, W, H, device, sh_degree
        self.gt_scene = syn.make_scene(gt_gaussians, W, H, sh_degree=sh_degree, seed=seed, sigma_px=4.0, frac_behind=0.0)
        g = torch.Generator().manual_seed(seed + 1000)
        axis = torch.tensor([0.1, 1.0, 0.05]); axis = axis / axis.norm()
        K = torch.tensor([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        self.w2c = []
        for f in range(n_frames):
            ang = step_angle * f
            R = torch.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * (K @ K)
            t = torch.tensor([step_shift * f, 0.2 * step_shift * math.sin(0.7 * f), 0.0]) + 0.002 * torch.randn(3, generator=g)
            M = torch.eye(4); M[:3, :3] = R; M[:3, 3] = t
            self.w2c.append(M)
        self.w2c = torch.stack(self.w2c)
        self._targets = {}
        self._gt_params = None
        self.pose_table = None      # stage A's result (`pose_dict`), when it was run: rel_pose() then answers from it

    # ---- cameras ---------------------------------------------------------------------------------------------------
    def settings_for_pose(self, pose_w2c: torch.Tensor, bg=None) -> GaussianRasterizationSettings:
        """Raster settings of a camera whose world-to-camera transform is `pose_w2c` ([4,4], in whatever coordinate
        system the model being rendered lives in)."""
        p = pose_w2c.detach().float().cpu()
        cam = syn.make_camera(self.W, self.H, R=p[:3, :3], t=p[:3, 3])
        d = self.device
        return GaussianRasterizationSettings(
            image_height=self.H, image_width=self.W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
            bg=(torch.zeros(3) if bg is None else bg).to(d), scale_modifier=1.0, viewmatrix=cam["viewmatrix"].to(d),
            projmatrix=cam["projmatrix"].to(d), sh_degree=self.sh_degree, campos=cam["campos"].to(d), prefiltered=False,
            debug=False)

    def true_rel_pose(self, a: int, b: int) -> torch.Tensor:
        """Ground-truth `rel_pose_{a}_to_{b}` (camera a coordinates -> camera b coordinates): what stage A estimates."""
        return self.w2c[b] @ torch.linalg.inv(self.w2c[a])

    def use_pose_table(self, pose_dict):
        """Adopt stage A's `pose_dict` (stage_a.run_stage_a): from here on the segments chain the ESTIMATED relative poses
        (ht3dgs_trainer.py:739-741, :783-790), as the reference does."""
        self.pose_table = {k: v.detach().float().cpu() for k, v in pose_dict.items()}

    def rel_pose(self, a: int, b: int) -> torch.Tensor:
        """`rel_pose_{a}_to_{b}`: stage A's estimate when one was adopted, else the ground truth standing in for it."""
        if self.pose_table is not None:
            m = self.pose_table.get(f"rel_pose_{a}_to_{b}")
            if m is not None:
                return m.clone()
        return self.true_rel_pose(a, b)

    # ---- targets ---------------------------------------------------------------------------------------------------
    def target(self, f: int) -> torch.Tensor:
        """Ground-truth image of frame f (rendered once, kept on the device)."""
        if f not in self._targets:
            from . import train_step as ts
            if self._gt_params is None:
                self._gt_params = ts.GaussianParams(self.gt_scene, self.device, optimizer="torch")
            with torch.no_grad():
                self._targets[f] = ts.render(self._gt_params, self.settings_for_pose(self.w2c[f]))["image"].clone()
        return self._targets[f]

    # ---- leaf initialisation -----------------------------------------------------------------------------------------
    def depth(self, f: int) -> torch.Tensor:
        """Expected depth map of frame f, [H, W] (stands in for the monocular depth the reference predicts per frame)."""
        from . import train_step as ts
        cache = self.__dict__.setdefault("_depths", {})
        if f in cache:
            return cache[f]
        if self._gt_params is None:
            self._gt_params = ts.GaussianParams(self.gt_scene, self.device, optimizer="torch")
        with torch.no_grad():
            pkg = ts.render(self._gt_params, self.settings_for_pose(self.w2c[f]))
            cache[f] = (pkg["depth"][0] / pkg["alpha"][0].clamp_min(1e-3)).clone()
        return cache[f]

    def pixel_scene(self, f: int, stride: int = 2, seed: int = 0, depth_noise: float = 0.02) -> Dict:
        """One Gaussian per `stride`-th pixel of frame f, un-projected with the frame's depth into that frame's camera
        coordinates, coloured with the pixel (SH band 0), sized to its pixel footprint -- the single-image initialisation of
        stage A (`init_model` from the un-projected monocular depth, ht3dgs_trainer.py:352-363, :172-212).  Such a model
        explains its image from the first iteration, which is what makes the photometric pose fit on the NEXT frame
        well-posed (a sparse subset of the scene is not: tools/stage_a_probe.py)."""
        g = torch.Generator().manual_seed(seed + 7919 * f)
        img = self.target(f).cpu()                       # [3, H, W]
        dep = self.depth(f).cpu()
        cam = syn.make_camera(self.W, self.H)
        ys, xs = torch.meshgrid(torch.arange(stride // 2, self.H, stride), torch.arange(stride // 2, self.W, stride), indexing="ij")
        ys, xs = ys.reshape(-1), xs.reshape(-1)
        z = dep[ys, xs]
        ok = z > 0.21                                     # in front of the near cut of the rasterizer
        ys, xs, z = ys[ok], xs[ok], z[ok]
        z = z * (1.0 + depth_noise * torch.randn(z.shape, generator=g))
        x = (xs.float() + 0.5 - 0.5 * self.W) / cam["fx"] * z
        y = (ys.float() + 0.5 - 0.5 * self.H) / cam["fy"] * z
        n = z.shape[0]
        sc = dict(self.gt_scene)
        sc["means3D"] = torch.stack((x, y, z), 1).float().contiguous()
        sc["scales"] = (0.7 * stride * z / cam["fx"])[:, None].repeat(1, 3).float().contiguous()
        rot = torch.zeros(n, 4); rot[:, 0] = 1.0
        sc["rotations"] = rot
        sc["opacities"] = torch.full((n, 1), 0.7)
        shs = torch.zeros(n, 16, 3)
        shs[:, 0] = (img[:, ys, xs].t() - 0.5) / 0.28209479177387814
        sc["shs"] = shs.contiguous()
        return sc

    def leaf_scene(self, start_fidx: int, n_points: int, seed: int, noise: float = 1.0) -> Dict:
        """A perturbed subset of the ground truth, expressed in the camera frame of `start_fidx` (stands in for
        `init_leaf_3DGS` from monocular depth, :172-212): positions jittered, colours / opacities / scales off."""
        g = torch.Generator().manual_seed(seed)
        gt = self.gt_scene
        n_gt = gt["means3D"].shape[0]
        idx = torch.randperm(n_gt, generator=g)[:min(n_points, n_gt)]
        if n_points > n_gt:
            idx = torch.cat((idx, torch.randint(0, n_gt, (n_points - n_gt,), generator=g)))
        M = self.w2c[start_fidx]
        xyz = gt["means3D"][idx] @ M[:3, :3].t() + M[:3, 3]
        # the Gaussians' own rotations are composed with the camera rotation: q' = q_M * q
        R = M[:3, :3]
        tr = R.trace()
        qw = math.sqrt(max(1e-12, 1.0 + float(tr))) / 2
        qM = torch.tensor([qw, float(R[2, 1] - R[1, 2]) / (4 * qw), float(R[0, 2] - R[2, 0]) / (4 * qw), float(R[1, 0] - R[0, 1]) / (4 * qw)])
        q = gt["rotations"][idx]
        w0, x0, y0, z0 = qM
        w1, x1, y1, z1 = q.unbind(1)
        rot = torch.stack((w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1, w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1,
                           w0 * y1 - x0 * z1 + y0 * w1 + z0 * x1, w0 * z1 + x0 * y1 - y0 * x1 + z0 * w1), dim=1)
        n = idx.shape[0]
        sc = dict(gt)
        sc["means3D"] = (xyz + noise * 0.01 * torch.randn(n, 3, generator=g)).float().contiguous()
        sc["rotations"] = rot.float().contiguous()
        sc["scales"] = (gt["scales"][idx] * torch.exp(noise * 0.2 * torch.randn(n, 3, generator=g))).contiguous()
        sc["opacities"] = (gt["opacities"][idx] * (1 - noise * 0.5 * torch.rand(n, 1, generator=g))).clamp(0.02, 0.98).contiguous()
        sc["shs"] = (gt["shs"][idx] + noise * 0.2 * torch.randn(n, 16, 3, generator=g) * (torch.arange(16) == 0).float()[None, :, None]).contiguous()
        return sc
