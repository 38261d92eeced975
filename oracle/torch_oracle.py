"""Dense float64 PyTorch-autograd restatement of the rasterizer -- TEST INFRASTRUCTURE ONLY.

Second, independent oracle: the forward is written as whole-image tensor algebra and the backward
comes from autograd, so it cross-checks the hand-derived backward of oracle/gsr_oracle.c.  Memory is
O(pixels x visible Gaussians): small cases only (<= ~64x64 px, <= ~2k Gaussians).

PARITY UNPINNED (see gsr_oracle.c header): the arithmetic restated here is the published algorithm
of the reference's un-vendored submodule (/root/reference/.gitmodules:4-6).  The reference-owned
pieces are followed literally: SH basis utils/sh_utils.py:57-100 (+0.5 / clamp_min
gaussian_model_ht.py:859-862), Sigma = (R S)(R S)^T utils/general_utils.py:76-108, matrices read
linearly as column-major scene/cameras.py:76-98.

Three places where the public module's backward is NOT the autograd derivative of its forward are
reproduced with explicit straight-through constructions so that both oracles define the same
gradient: (1) min(0.99, alpha) passes gradient through, (2) the frustum-clamped view coordinate is
a constant in d/dz, (3) the conic inverse uses 1/(det^2 + 1e-7) in the backward.
"""
import math

import numpy as np
import torch

TILE = 16
C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
      -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def sh_basis_eval(deg, sh, d):
    """sh [V,M,3], d [V,3] unit -> [V,3]; polynomial of utils/sh_utils.py:74-100."""
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    res = C0 * sh[:, 0]
    if deg > 0:
        res = res - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6]
               + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        res = (res + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10]
               + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
               + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14]
               + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res


class _ConicInv(torch.autograd.Function):
    """(a,b,c) -> (c,-b,a)/det with the module's guarded backward 1/(det^2+1e-7)."""

    @staticmethod
    def forward(ctx, a, b, c):
        det = a * c - b * b
        ctx.save_for_backward(a, b, c)
        return c / det, -b / det, a / det

    @staticmethod
    def backward(ctx, gA, gB, gC):
        a, b, c = ctx.saved_tensors
        det = a * c - b * b
        d2i = 1.0 / (det * det + 1e-7)
        ga = d2i * (-c * c * gA + b * c * gB - b * b * gC)
        gb = d2i * (2 * b * c * gA - (det + 2 * b * b) * gB + 2 * a * b * gC)
        gc = d2i * (-b * b * gA + a * b * gB - a * a * gC)
        return ga, gb, gc


def depth_key32(vm32, p32):
    """binary32 fmaf chain identical to gsr_oracle.c:depth_key (emulated exactly in float64)."""
    vm = vm32.astype(np.float64)
    p = p32.astype(np.float64)

    def fma32(a, b, c):  # a*b+c is exact in f64 for f32 operands up to one rounding; round once to f32
        return (a * b + c).astype(np.float32).astype(np.float64)

    t = fma32(vm[2], p[:, 0], vm[14])
    t = fma32(vm[6], p[:, 1], t)
    t = fma32(vm[10], p[:, 2], t)
    return t.astype(np.float32)


def render(means3D, opacities, viewmatrix, projmatrix, campos, bg, image_height, image_width,
           tanfovx, tanfovy, sh_degree=0, shs=None, colors_precomp=None, scales=None, rotations=None,
           cov3D_precomp=None, scale_modifier=1.0, means2D=None):
    """All tensor inputs float64 (leaf tensors may require grad).  Returns color, radii, depth, alpha.
    means2D (zeros [N,3]) is the module's gradient sink: it enters as an NDC offset so that its grad is
    d/d(ndc) = d/d(pixel) * (W/2, H/2) (gaussian_model_ht.py:791-803, :718-721)."""
    dt = torch.float64
    N = means3D.shape[0]
    W, H = int(image_width), int(image_height)
    vm = viewmatrix.reshape(16).to(dt)
    pm = projmatrix.reshape(16).to(dt)
    X, Y, Z = means3D[:, 0], means3D[:, 1], means3D[:, 2]
    key = torch.from_numpy(depth_key32(vm.detach().numpy().astype(np.float32),
                                       means3D.detach().numpy().astype(np.float32)).astype(np.float64))
    t2 = vm[2] * X + vm[6] * Y + vm[10] * Z + vm[14]
    depth = t2 + (key - t2).detach()
    front = key > 0.2
    hx = pm[0] * X + pm[4] * Y + pm[8] * Z + pm[12]
    hy = pm[1] * X + pm[5] * Y + pm[9] * Z + pm[13]
    hw = pm[3] * X + pm[7] * Y + pm[11] * Z + pm[15]
    pw = 1.0 / (hw + 1e-7)
    ndcx, ndcy = hx * pw, hy * pw
    if means2D is not None:
        ndcx, ndcy = ndcx + means2D[:, 0], ndcy + means2D[:, 1]
    px = ((ndcx + 1) * W - 1) * 0.5
    py = ((ndcy + 1) * H - 1) * 0.5

    if cov3D_precomp is not None:
        S = cov3D_precomp
        Sig = torch.stack([S[:, 0], S[:, 1], S[:, 2], S[:, 1], S[:, 3], S[:, 4], S[:, 2], S[:, 4], S[:, 5]], 1).view(N, 3, 3)
    else:
        r, x, y, z = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                         2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                         2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).view(N, 3, 3)
        L = R * (scale_modifier * scales)[:, None, :]
        Sig = L @ L.transpose(1, 2)

    t0 = vm[0] * X + vm[4] * Y + vm[8] * Z + vm[12]
    t1 = vm[1] * X + vm[5] * Y + vm[9] * Z + vm[13]
    t2s = torch.where(front, t2, torch.ones_like(t2))  # keep culled rows finite
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz, tytz = t0 / t2s, t1 / t2s
    cx = txtz.clamp(-limx, limx)
    cy = tytz.clamp(-limy, limy)
    t0c = torch.where((txtz < -limx) | (txtz > limx), (cx * t2s).detach(), t0)
    t1c = torch.where((tytz < -limy) | (tytz > limy), (cy * t2s).detach(), t1)
    fx, fy = W / (2 * tanfovx), H / (2 * tanfovy)
    J00, J02 = fx / t2s, -fx * t0c / (t2s * t2s)
    J11, J12 = fy / t2s, -fy * t1c / (t2s * t2s)
    Wr = vm.view(4, 4).t()[:3, :3]  # Wr[r,k] = vm[k*4+r]
    m0 = J00[:, None] * Wr[0][None, :] + J02[:, None] * Wr[2][None, :]
    m1 = J11[:, None] * Wr[1][None, :] + J12[:, None] * Wr[2][None, :]
    Sm0 = (Sig @ m0[:, :, None])[:, :, 0]
    Sm1 = (Sig @ m1[:, :, None])[:, :, 0]
    a = (m0 * Sm0).sum(1) + 0.3
    b = (m0 * Sm1).sum(1)
    c = (m1 * Sm1).sum(1) + 0.3
    det = a * c - b * b
    A_, B_, C_ = _ConicInv.apply(a, b, c)
    mid = 0.5 * (a + c)
    disc = (mid * mid - det).clamp_min(0.1)
    lam = torch.maximum(mid + disc.sqrt(), mid - disc.sqrt())
    rad = torch.ceil(3 * lam.sqrt()).detach()
    tiles_x, tiles_y = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE

    def _clampi(v, hi):
        return torch.minimum(torch.maximum(torch.trunc(v), torch.zeros_like(v)), torch.full_like(v, hi))

    pxd, pyd = px.detach(), py.detach()
    x0 = _clampi((pxd - rad) / TILE, tiles_x)
    y0 = _clampi((pyd - rad) / TILE, tiles_y)
    x1 = _clampi((pxd + rad + TILE - 1) / TILE, tiles_x)
    y1 = _clampi((pyd + rad + TILE - 1) / TILE, tiles_y)
    vis = front & (det != 0) & (((x1 - x0) * (y1 - y0)) > 0)
    radii = torch.where(vis, rad, torch.zeros_like(rad)).to(torch.int32)

    if colors_precomp is not None:
        rgb = colors_precomp
    else:
        d = means3D - campos.to(dt)[None, :]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(sh_basis_eval(sh_degree, shs, d) + 0.5, 0.0)

    # ---- global blend order: (depth bits, index); per pixel only Gaussians whose rect covers its tile
    idx = torch.nonzero(vis)[:, 0]
    order = np.lexsort((idx.numpy(), key[idx].numpy().astype(np.float32).view(np.uint32)))
    idx = idx[torch.from_numpy(order)]
    V = idx.numel()
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    pix_x, pix_y = xs.reshape(-1, 1), ys.reshape(-1, 1)
    tile_x, tile_y = torch.floor(pix_x / TILE), torch.floor(pix_y / TILE)
    if V == 0:
        zero = (means3D.sum() + opacities.sum()) * 0
        color = bg.to(dt).view(3, 1, 1).expand(3, H, W) + zero
        return color, radii, torch.zeros(1, H, W, dtype=dt) + zero, torch.zeros(1, H, W, dtype=dt) + zero
    in_rect = ((tile_x >= x0[idx][None]) & (tile_x < x1[idx][None]) & (tile_y >= y0[idx][None]) & (tile_y < y1[idx][None]))
    dx = px[idx][None, :] - pix_x
    dy = py[idx][None, :] - pix_y
    power = -0.5 * (A_[idx][None] * dx * dx + C_[idx][None] * dy * dy) - B_[idx][None] * dx * dy
    o = opacities.reshape(-1)[idx][None]
    a_raw = o * torch.exp(power.clamp_max(0.0))
    alpha = a_raw + (a_raw.clamp_max(0.99) - a_raw).detach()  # straight-through min(0.99, .)
    valid = in_rect & (power <= 0) & (alpha >= 1.0 / 255.0)
    a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
    one_m = 1 - a_eff
    T_incl = torch.cumprod(one_m, dim=1)
    T_excl = torch.cat([torch.ones_like(T_incl[:, :1]), T_incl[:, :-1]], 1)
    stop = valid & (T_incl < 1e-4)
    seen_stop = torch.cumsum(stop.to(torch.int64), 1) > 0
    live = valid & ~seen_stop
    w = torch.where(live, a_eff * T_excl, torch.zeros_like(a_eff))
    T_final = torch.prod(torch.where(live, one_m, torch.ones_like(one_m)), dim=1)
    C = w @ rgb[idx]
    D = w @ depth[idx]
    A = w.sum(1)
    color = (C + T_final[:, None] * bg.to(dt)[None, :]).t().reshape(3, H, W)
    return color, radii, D.reshape(1, H, W), A.reshape(1, H, W)


def render_from_f32(inputs, grads_out=None):
    """inputs: dict of float32 numpy arrays (+ scalars).  Returns outputs and (optionally) input grads."""
    names = ["means3D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp"]
    t = {}
    for k in names:
        v = inputs.get(k)
        t[k] = None if v is None else torch.tensor(np.asarray(v, np.float64), requires_grad=True)
    t["means2D"] = torch.zeros(t["means3D"].shape[0], 3, dtype=torch.float64, requires_grad=True)
    const = {k: torch.tensor(np.asarray(inputs[k], np.float64), requires_grad=(k != "bg"))
             for k in ["viewmatrix", "projmatrix", "campos", "bg"]}
    color, radii, depth, alpha = render(
        t["means3D"], t["opacities"], const["viewmatrix"], const["projmatrix"], const["campos"], const["bg"],
        inputs["image_height"], inputs["image_width"], inputs["tanfovx"], inputs["tanfovy"],
        sh_degree=inputs.get("sh_degree", 0), shs=t["shs"], colors_precomp=t["colors_precomp"],
        scales=t["scales"], rotations=t["rotations"], cov3D_precomp=t["cov3D_precomp"],
        scale_modifier=inputs.get("scale_modifier", 1.0), means2D=t["means2D"])
    out = dict(color=color.detach().numpy(), radii=radii.numpy(), depth=depth.detach().numpy(), alpha=alpha.detach().numpy())
    if grads_out is not None:
        gc, gd, ga = (torch.tensor(np.asarray(g, np.float64)) if g is not None else None for g in grads_out)
        loss = (color * gc).sum()
        if gd is not None:
            loss = loss + (depth * gd.reshape(depth.shape)).sum()
        if ga is not None:
            loss = loss + (alpha * ga.reshape(alpha.shape)).sum()
        leaves = [(k, v) for k, v in t.items() if v is not None] + [(k, const[k]) for k in ("viewmatrix", "projmatrix", "campos")]
        gs = torch.autograd.grad(loss, [v for _, v in leaves], allow_unused=True)
        out["grads"] = {k: (g.numpy() if g is not None else np.zeros(tuple(v.shape))) for (k, v), g in zip(leaves, gs)}
    return out
