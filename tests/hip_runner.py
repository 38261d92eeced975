"""Run the product path (diff_gaussian_rasterization -> C ABI -> HIP kernels) on a scene dict; GPU only."""
import numpy as np
import torch

from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

GRAD_KEYS = ["means3D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp"]


def settings_from(kw, dev, noncontig=False, cam_grad=False):
    vm, pm, cp = kw["viewmatrix"].to(dev), kw["projmatrix"].to(dev), kw["campos"].to(dev)
    if cam_grad:
        vm, pm, cp = vm.clone().requires_grad_(True), pm.clone().requires_grad_(True), cp.clone().requires_grad_(True)
    if noncontig:  # the reference hands a transpose view and a row slice
        vm = vm.t().contiguous().t()
        cp = torch.stack([cp, cp], 1)[:, 0]
        assert not vm.is_contiguous()
    return GaussianRasterizationSettings(
        image_height=int(kw["image_height"]), image_width=int(kw["image_width"]), tanfovx=float(kw["tanfovx"]),
        tanfovy=float(kw["tanfovy"]), bg=kw["bg"].to(dev), scale_modifier=float(kw.get("scale_modifier", 1.0)),
        viewmatrix=vm, projmatrix=pm, sh_degree=int(kw.get("sh_degree", 0)), campos=cp, prefiltered=False, debug=False)


def run_hip(kw, grads=None, dev="cuda:0", noncontig=False, cam_grad=False):
    dev = torch.device(dev)
    t = {}
    for k in GRAD_KEYS:
        v = kw.get(k)
        t[k] = None if v is None else v.detach().to(dev).float().requires_grad_(grads is not None)
    N = t["means3D"].shape[0]
    m2d = torch.zeros(N, 3, device=dev, requires_grad=grads is not None)
    rs = settings_from(kw, dev, noncontig, cam_grad and grads is not None)
    rast = GaussianRasterizer(rs)
    color, radii, depth, alpha = rast(means3D=t["means3D"], means2D=m2d, shs=t["shs"], colors_precomp=t["colors_precomp"],
                                      opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"],
                                      cov3D_precomp=t["cov3D_precomp"])
    out = dict(fwd=(color.detach().cpu().numpy(), radii.cpu().numpy(), depth.detach().cpu().numpy(),
                    alpha.detach().cpu().numpy()))
    if grads is not None:
        gc, gd, ga = grads
        loss = (color * torch.from_numpy(gc).to(dev)).sum()
        if gd is not None:
            loss = loss + (depth[0] * torch.from_numpy(gd).to(dev)).sum()
        if ga is not None:
            loss = loss + (alpha[0] * torch.from_numpy(ga).to(dev)).sum()
        loss.backward()
        def _g(v):
            return (v.grad if v.grad is not None else torch.zeros_like(v)).detach().cpu().numpy()
        g = {k: _g(v) for k, v in t.items() if v is not None}
        g["means2D"] = _g(m2d)
        if cam_grad:
            g["viewmatrix"], g["projmatrix"], g["campos"] = _g(rs.viewmatrix), _g(rs.projmatrix), _g(rs.campos)
        out["grads"] = g
    return out
