"""Host-side mirror of the reference's rasterizer interface, on top of the C ABI of include/gsr.h.

Same names, argument meaning and error behaviour as the module the reference imports
(`from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer`:
/root/reference/scene/gaussian_model_ht.py:39-42, gaussian_renderer/__init__.py:12), called as at
gaussian_model_ht.py:809-880 and returning the 4-tuple unpacked at :881-894
(color[3,H,W], radii[N] int32, depth[1,H,W], alpha[1,H,W]).

PyTorch is plumbing only: device memory (caching allocator), the current HIP stream and autograd.  All
arithmetic happens in the hand-written HIP kernels of csrc/gsr_kernels.hip.  There is no CPU path: tensors
must live on a ROCm device and the libraries must be built, otherwise a RuntimeError is raised.

Binding: the PyTorch-ROCm C++ extension `torch.ops.gsr.rasterize` (csrc/torch_ext.cpp: C++ autograd function, buffers
from at::empty, no Python in the allocator path).  `GSR_BINDING=ctypes` selects the plain-FFI route over the same C ABI
(`_RasterizeGaussians` below: the example INTEGRATION.md walks through).
"""
import ctypes as C
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _ext as E
from . import _lib as L


class GaussianRasterizationSettings(NamedTuple):
    """The 12 keywords built at gaussian_model_ht.py:809-822 / gaussian_renderer/__init__.py:38-51."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


# diagnostics of the most recent forward (bench.py reads R and the per-tile staged counters from here)
_LAST = {"num_rendered": 0, "image": None, "W": 0, "H": 0, "B": 1}


def _sync_last():
    """Pull the extension's record of the most recent forward into _LAST (the ctypes route fills _LAST itself)."""
    if E.use_ctypes() or not E._loaded:
        return
    rec = torch.ops.gsr.debug_last()
    if len(rec) == 4:
        image, binning, meta, dims = rec
        _LAST.update(num_rendered=int(meta[0]), binning_capacity=int(meta[1]), image=image, binning=binning if binning.numel() else None,
                     W=int(dims[0]), H=int(dims[1]), B=int(dims[2]) if dims.numel() > 2 else 1)


def last_call_info():
    """{'num_rendered': R, 'staged': R_eff} of the most recent forward on this process."""
    lib = L.load()
    _sync_last()
    img, W, H, B = _LAST["image"], _LAST["W"], _LAST["H"], _LAST.get("B", 1)
    staged = 0
    if img is not None:
        T = ((W + 15) // 16) * ((H + 15) // 16) * B          # (a batched render: B images, one tall tile grid)
        off = lib.gsr_image_bytes_batched(W, H, B) - ((T * 16 + 255) // 256) * 256
        # four counters per tile (one per 8x8 sub-tile wave; tile-level kernels use slot 0): the tile's staged depth
        # is the deepest of its waves
        staged = int(img[off:off + 16 * T].view(torch.int32).view(T, 4).max(dim=1).values.sum().item())
    return {"num_rendered": _LAST["num_rendered"], "staged": staged}


def last_binning():
    """Debug / test hook (gsr_debug_read_binning): (ranges[T,2] int64, list[R] int64) of the most recent forward on this
    process -- per-tile [begin, end) into the (tile, depth, id)-ordered list of Gaussian ids."""
    lib = L.load()
    _sync_last()
    b, W, H, R = _LAST.get("binning"), _LAST["W"], _LAST["H"], _LAST["num_rendered"]
    if b is None:
        raise RuntimeError("no forward has run yet")
    if _LAST.get("B", 1) > 1:
        raise RuntimeError("last_binning: not available for a batched render")
    T = ((W + 15) // 16) * ((H + 15) // 16)
    ranges = torch.empty((T, 2), dtype=torch.int32, device=b.device)
    lst = torch.empty((max(R, 1),), dtype=torch.int32, device=b.device)
    with torch.cuda.device(b.device):
        st = torch.cuda.current_stream(b.device).cuda_stream
        L.check(lib.gsr_debug_read_binning(b.data_ptr(), _LAST["binning_capacity"], R, W, H, ranges.data_ptr(), lst.data_ptr(),
                                           C.c_void_p(st)), "gsr_debug_read_binning")
    return ranges.long() & 0xffffffff, lst[:R].long() & 0xffffffff


def _f32c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """float32 + contiguous (viewmatrix / campos arrive as non-contiguous views: SURVEY Appendix B)."""
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return None if (t is None or t.numel() == 0) else t.data_ptr()


def _empty_to_none(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if (t is None or t.numel() == 0) else t


class _Workspace:
    """Allocator handed to gsr_forward: R-sized buffers come from torch's caching allocator."""

    def __init__(self, device):
        self.device = device
        self.binning = None
        self.scratch = []
        self.cb = L.ALLOC_FN(self._alloc)

    def _alloc(self, nbytes, tag, _user):
        try:
            buf = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        except Exception:  # out of memory -> NULL -> GSR_ERR_ALLOC
            return None
        if tag == L.GSR_ALLOC_BINNING:
            self.binning = buf
        else:
            self.scratch.append(buf)
        return buf.data_ptr()


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                sh_rest=None, raw_params=False, viewmatrix=None, projmatrix=None, campos=None, fused_adam=None,
                points_transform=None, view_id=0):
        # viewmatrix / projmatrix / campos are ALSO passed as explicit tensor inputs (same objects as in
        # raster_settings) so that autograd can return their gradients: a NamedTuple cannot carry grads.
        lib = L.load()
        rs = raster_settings
        dev = means3D.device
        if dev.type != "cuda":
            raise RuntimeError("GaussianRasterizer: tensors must be on a ROCm/HIP device (no CPU fallback)")
        means3D = _f32c(means3D)
        N = means3D.shape[0]
        sh, colors_precomp = _f32c(_empty_to_none(sh)), _f32c(_empty_to_none(colors_precomp))
        scales, rotations = _f32c(_empty_to_none(scales)), _f32c(_empty_to_none(rotations))
        cov3Ds_precomp = _f32c(_empty_to_none(cov3Ds_precomp))
        opacities = _f32c(opacities)
        vm = _f32c((viewmatrix if viewmatrix is not None else rs.viewmatrix).to(dev))
        pm = _f32c((projmatrix if projmatrix is not None else rs.projmatrix).to(dev))
        campos = _f32c((campos if campos is not None else rs.campos).to(dev))
        bg = _f32c(rs.bg.to(dev))
        H, W = int(rs.image_height), int(rs.image_width)
        sh_rest = _f32c(_empty_to_none(sh_rest))
        xf = None
        if points_transform is not None:       # [3,4] or [4,4] rigid / affine transform applied to the means in-kernel
            if tuple(points_transform.shape) not in ((3, 4), (4, 4)):
                raise RuntimeError("points_transform must be a [3,4] or [4,4] tensor")
            xf = _f32c(points_transform.to(dev)[:3])
        M = (int(sh.shape[1]) + (int(sh_rest.shape[1]) if sh_rest is not None else 0)) if sh is not None else 0

        color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        depth = torch.empty((1, H, W), dtype=torch.float32, device=dev)
        alpha = torch.empty((1, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((N,), dtype=torch.int32, device=dev)
        geom = torch.empty(lib.gsr_geom_bytes(N), dtype=torch.uint8, device=dev)
        image = torch.empty(lib.gsr_image_bytes(W, H), dtype=torch.uint8, device=dev)
        ws = _Workspace(dev)

        a = L.GsrForwardArgs()
        a.N, a.M, a.D, a.W, a.H = N, M, int(rs.sh_degree), W, H
        a.prefiltered, a.debug = int(bool(rs.prefiltered)), int(bool(rs.debug))
        a.scale_modifier, a.tanfovx, a.tanfovy = float(rs.scale_modifier), float(rs.tanfovx), float(rs.tanfovy)
        a.means3D, a.scales, a.rotations, a.cov3D_precomp = _ptr(means3D), _ptr(scales), _ptr(rotations), _ptr(cov3Ds_precomp)
        a.opacities, a.shs, a.colors_precomp = _ptr(opacities), _ptr(sh), _ptr(colors_precomp)
        a.viewmatrix, a.projmatrix, a.campos, a.bg = _ptr(vm), _ptr(pm), _ptr(campos), _ptr(bg)
        a.out_color, a.out_depth, a.out_alpha, a.radii = color.data_ptr(), depth.data_ptr(), alpha.data_ptr(), _ptr(radii)
        a.geom, a.image = geom.data_ptr(), image.data_ptr()
        a.alloc, a.alloc_user = ws.cb, None
        a.shs_rest, a.raw_params = _ptr(sh_rest), int(bool(raw_params))
        a.points_transform = _ptr(xf)
        a.view_id = int(view_id)
        out = L.GsrForwardOut()
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            L.check(lib.gsr_forward(C.byref(a), C.byref(out), C.c_void_p(stream)), "gsr_forward")
        ws.scratch.clear()  # stream-ordered reuse by the caching allocator is safe: same stream

        _LAST.update(num_rendered=int(out.num_rendered), image=image, W=W, H=H, B=1, binning=ws.binning,
                     binning_capacity=int(out.binning_capacity))
        ctx.raster_settings = rs
        ctx.num_rendered = int(out.num_rendered)
        ctx.binning_capacity = int(out.binning_capacity)
        ctx.forward_flags = int(out.forward_flags)
        ctx.dims = (N, M, H, W)
        ctx.has = (sh is not None, colors_precomp is not None, scales is not None, cov3Ds_precomp is not None)
        ctx.raw = (sh_rest is not None, bool(raw_params))
        ctx.fused_adam = fused_adam
        ctx.xf_shape = tuple(points_transform.shape) if points_transform is not None else None
        z = means3D.new_empty(0)
        # NOTE: depth is deliberately NOT saved -- the caller mutates it in place (ht3dgs_trainer.py:1290-1292).
        ctx.save_for_backward(means3D, opacities, sh if sh is not None else z, colors_precomp if colors_precomp is not None else z,
                              scales if scales is not None else z, rotations if rotations is not None else z,
                              cov3Ds_precomp if cov3Ds_precomp is not None else z, vm, pm, campos, bg, geom, image,
                              ws.binning, sh_rest if sh_rest is not None else z, xf if xf is not None else z)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)   # unused depth / alpha outputs arrive as None -> specialised backward
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        lib = L.load()
        rs = ctx.raster_settings
        (means3D, opacities, sh, colors_precomp, scales, rotations, cov3Ds_precomp, vm, pm, campos, bg, geom, image,
         binning, sh_rest, xf) = ctx.saved_tensors
        has_rest, raw_params = ctx.raw
        N, M, H, W = ctx.dims
        has_sh, has_col, has_scale, has_cov = ctx.has
        dev = means3D.device
        grad_color, grad_depth, grad_alpha = _f32c(grad_color), _f32c(grad_depth), _f32c(grad_alpha)
        if grad_color is None and grad_depth is None and grad_alpha is None:
            return (None,) * 17
        need_vm, need_pm, need_cp = ctx.needs_input_grad[11], ctx.needs_input_grad[12], ctx.needs_input_grad[13]
        d_vm = torch.empty((4, 4), dtype=torch.float32, device=dev) if need_vm else None
        d_pm = torch.empty((4, 4), dtype=torch.float32, device=dev) if need_pm else None
        d_cp = torch.empty((3,), dtype=torch.float32, device=dev) if need_cp else None
        has_xf = ctx.xf_shape is not None
        d_xf = torch.zeros(ctx.xf_shape, dtype=torch.float32, device=dev) if (has_xf and ctx.needs_input_grad[15]) else None

        fused = ctx.fused_adam
        d_means2D = torch.empty((N, 3), dtype=torch.float32, device=dev)
        d_means3D = d_opac = d_sh = d_sh_rest = d_col = d_scales = d_rot = d_cov = None
        fa = None
        if fused is not None:
            # optimizer-in-backward: the kernel applies the Adam step to the raw parameters in place; their gradients
            # are never materialised (the corresponding .grad stay None and optimizer.step() has nothing left to do)
            if not (raw_params and has_rest and has_sh and has_scale):
                raise RuntimeError("fused_adam needs the raw-parameter path (rasterize_gaussians_raw)")
            fa = fused.fused_backward_args({"xyz": means3D, "f_dc": sh, "f_rest": sh_rest, "opacity": opacities,
                                            "scaling": scales, "rotation": rotations}, int(rs.sh_degree))
        else:
            d_means3D = torch.empty((N, 3), dtype=torch.float32, device=dev)
            d_opac = torch.empty((N, 1), dtype=torch.float32, device=dev)
            d_sh = torch.empty((N, 1 if has_rest else M, 3), dtype=torch.float32, device=dev) if has_sh else None
            d_sh_rest = torch.empty((N, M - 1, 3), dtype=torch.float32, device=dev) if (has_sh and has_rest) else None
            d_col = torch.empty((N, 3), dtype=torch.float32, device=dev) if has_col else None
            d_scales = torch.empty((N, 3), dtype=torch.float32, device=dev) if has_scale else None
            d_rot = torch.empty((N, 4), dtype=torch.float32, device=dev) if has_scale else None
            d_cov = torch.empty((N, 6), dtype=torch.float32, device=dev) if has_cov else None
        scratch = torch.empty(lib.gsr_backward_scratch_bytes(N), dtype=torch.uint8, device=dev)

        a = L.GsrBackwardArgs()
        a.N, a.M, a.D, a.W, a.H = N, M, int(rs.sh_degree), W, H
        a.scale_modifier, a.tanfovx, a.tanfovy = float(rs.scale_modifier), float(rs.tanfovx), float(rs.tanfovy)
        a.means3D, a.opacities = _ptr(means3D), _ptr(opacities)
        a.scales, a.rotations = (_ptr(scales), _ptr(rotations)) if has_scale else (None, None)
        a.cov3D_precomp = _ptr(cov3Ds_precomp) if has_cov else None
        a.shs = _ptr(sh) if has_sh else None
        a.colors_precomp = _ptr(colors_precomp) if has_col else None
        a.viewmatrix, a.projmatrix, a.campos, a.bg = _ptr(vm), _ptr(pm), _ptr(campos), _ptr(bg)
        a.geom, a.image, a.binning, a.num_rendered = geom.data_ptr(), image.data_ptr(), binning.data_ptr(), ctx.num_rendered
        a.binning_capacity = ctx.binning_capacity
        a.forward_flags = ctx.forward_flags
        a.grad_color, a.grad_depth, a.grad_alpha = _ptr(grad_color), _ptr(grad_depth), _ptr(grad_alpha)
        a.d_means3D, a.d_means2D, a.d_opacities = _ptr(d_means3D), _ptr(d_means2D), _ptr(d_opac)
        a.d_colors_precomp, a.d_shs = _ptr(d_col), _ptr(d_sh)
        a.d_scales, a.d_rotations, a.d_cov3D_precomp = _ptr(d_scales), _ptr(d_rot), _ptr(d_cov)
        a.scratch = scratch.data_ptr()
        a.shs_rest = _ptr(sh_rest) if has_rest else None
        a.d_shs_rest, a.raw_params = _ptr(d_sh_rest), int(raw_params)
        a.d_viewmatrix, a.d_projmatrix, a.d_campos = _ptr(d_vm), _ptr(d_pm), _ptr(d_cp)
        a.fused_adam = C.addressof(fa) if fa is not None else None
        a.points_transform = _ptr(xf) if has_xf else None
        a.d_points_transform = _ptr(d_xf)       # first 12 floats = rows 0..2 (a [4,4] input keeps a zero last row)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            L.check(lib.gsr_backward(C.byref(a), C.c_void_p(stream)), "gsr_backward")
        if fused is not None:
            fused.fused_backward_applied()      # the update is enqueued: the optimizer's step count advances now, not at render time
        return (d_means3D, d_means2D, d_sh, d_col, d_opac, d_scales, d_rot, d_cov, None, d_sh_rest, None, d_vm, d_pm, d_cp, None, d_xf, None)


def _cam_inputs(rs):
    """Only route the camera tensors through autograd when one of them wants a gradient."""
    if any(torch.is_tensor(t) and t.requires_grad for t in (rs.viewmatrix, rs.projmatrix, rs.campos)):
        return rs.viewmatrix, rs.projmatrix, rs.campos
    return None, None, None


_EMPTY = {}


def _e(dev):
    t = _EMPTY.get(dev)
    if t is None:
        t = _EMPTY[dev] = torch.empty(0, dtype=torch.float32, device=dev)
    return t


def _ecpu():
    t = _EMPTY.get("cpu_i64")
    if t is None:
        t = _EMPTY["cpu_i64"] = torch.empty(0, dtype=torch.int64)
    return t


def _rasterize_ext(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs, sh_rest, raw_params,
                   fused_adam, points_transform, prepared=None, prepare_next=None, next_points_transform=None, densify_stats=None,
                   batch_first_block=None, fused_adam_deferred=False, view_id=0, extras=0):
    """torch.ops.gsr.rasterize: empty tensors stand for None; the camera tensors of the settings tuple are ordinary inputs
    (their gradients are produced when one of them requires grad)."""
    ops = E.load()
    dev = means3D.device
    if dev.type != "cuda":
        raise RuntimeError("GaussianRasterizer: tensors must be on a ROCm/HIP device (no CPU fallback)")
    e = _e(dev)
    pick = lambda t: e if (t is None or t.numel() == 0) else t
    vm, pm, cp = rs.viewmatrix, rs.projmatrix, rs.campos
    cam_grad = vm.requires_grad or pm.requires_grad or cp.requires_grad
    if vm.device != dev:
        vm, pm, cp = vm.to(dev), pm.to(dev), cp.to(dev)
    bg = rs.bg if rs.bg.device == dev else rs.bg.to(dev)
    nb = (len(batch_first_block) - 1) if batch_first_block is not None else 1
    if points_transform is not None and tuple(points_transform.shape) not in (((3, 4), (4, 4)) if nb <= 1 else ((nb, 3, 4),)):
        raise RuntimeError("points_transform must be a [3,4] or [4,4] tensor ([B,3,4] for a batch of B models)")
    if nb > 1 and (tuple(vm.shape) != (nb, 4, 4) or tuple(pm.shape) != (nb, 4, 4) or tuple(cp.shape) != (nb, 3)):
        raise RuntimeError("batch: raster_settings.viewmatrix / projmatrix must be [B,4,4] and campos [B,3]")
    xf = e if points_transform is None else points_transform.to(dev)
    m, v, lr, b1, b2, eps, step, commit = [], [], [], 0.0, 0.0, 0.0, 0, None
    if fused_adam is not None and not (torch.is_grad_enabled() and (means3D.requires_grad or opacities.requires_grad)):
        fused_adam = None       # a render that cannot reach a backward (torch.no_grad(), detached parameters): nothing to plan
    if fused_adam is not None:
        if not (raw_params and sh_rest is not None and sh is not None and scales is not None):
            raise RuntimeError("fused_adam needs the raw-parameter path (rasterize_gaussians_raw)")
        # a PLAN only: the step count advances when the backward that applies the update runs (csrc/torch_ext.cpp increments
        # `commit`), so a forward whose graph is dropped leaves the optimizer untouched
        group_tensors = {"xyz": means3D, "f_dc": sh, "f_rest": sh_rest, "opacity": opacities, "scaling": scales, "rotation": rotations}
        if fused_adam_deferred:      # the update goes to shadow buffers; optimizer.step() adopts it (optim.FusedAdam, deferred application)
            if prepare_next is not None:
                raise RuntimeError("fused_adam_deferred: a deferred update cannot prepare a next view")
            m, v, lr, b1, b2, eps, step, commit = fused_adam.deferred_step_plan(group_tensors, int(rs.sh_degree))
        else:
            m, v, lr, b1, b2, eps, step, commit = fused_adam.fused_step_plan(group_tensors, int(rs.sh_degree),
                                                                             None if prepare_next is None else int(prepare_next.sh_degree))
    eb = _EMPTY.get(("u8", dev))
    if eb is None:
        eb = _EMPTY[("u8", dev)] = torch.empty(0, dtype=torch.uint8, device=dev)
    nx = prepare_next
    if nx is not None and fused_adam is None and not torch.is_grad_enabled():
        nx = None               # (a no_grad render has no backward that could prepare anything)
    if nx is not None:
        if fused_adam is None:
            raise RuntimeError("prepare_next needs fused_adam: the backward that applies the update prepares the next render")
        if int(nx.sh_degree) not in (int(rs.sh_degree), int(rs.sh_degree) + 1) or int(nx.sh_degree) > 3 or \
                float(nx.scale_modifier) != float(rs.scale_modifier):
            raise RuntimeError("prepare_next: the next view must use this view's sh_degree or the one above it (oneupSHdegree) and this "
                               "view's scale_modifier")
        nvm, npm, ncp = nx.viewmatrix.to(dev), nx.projmatrix.to(dev), nx.campos.to(dev)
    args = (means3D, means2D, pick(sh), pick(colors_precomp), opacities, pick(scales), pick(rotations), pick(cov3Ds_precomp),
            pick(sh_rest), vm, pm, cp, bg, xf, int(rs.image_height), int(rs.image_width), float(rs.tanfovx),
            float(rs.tanfovy), float(rs.scale_modifier), int(rs.sh_degree), bool(raw_params), bool(rs.prefiltered),
            bool(rs.debug), bool(cam_grad), m, v, lr, b1, b2, eps, step,
            eb if prepared is None else prepared, e if nx is None else nvm, e if nx is None else npm, e if nx is None else ncp,
            0 if nx is None else int(nx.image_height), 0 if nx is None else int(nx.image_width),
            0.0 if nx is None else float(nx.tanfovx), 0.0 if nx is None else float(nx.tanfovy),
            e if (nx is None or next_points_transform is None) else next_points_transform.to(dev),
            -1 if nx is None else int(nx.sh_degree), _ecpu() if commit is None else commit,
            [] if densify_stats is None else list(densify_stats), [] if nb <= 1 else [int(x) for x in batch_first_block], int(view_id), int(extras))
    if not rs.debug:
        out = ops.rasterize(*args)
        if extras:       # (color, radii, depth, alpha, clamped colour, visibility bytes)
            return out[:4] + (out[5], out[6])
        return out[:5] if prepare_next is not None else out[:4]
    # raster_settings.debug = True: what the public module does -- on an error in the native forward, dump the arguments to
    # snapshot_fw.dump for offline inspection and re-raise (the reference always passes debug=False, gaussian_model_ht.py:821)
    try:
        out = ops.rasterize(*args)
        if extras:
            return out[:4] + (out[5], out[6])
        return out[:5] if prepare_next is not None else out[:4]
    except Exception:
        torch.save([a.detach().cpu() if torch.is_tensor(a) else a for a in args[:24]], "snapshot_fw.dump")
        print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
        raise


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                        points_transform=None, view_id=0):
    if not E.use_ctypes():
        return _rasterize_ext(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                              None, False, None, points_transform, view_id=view_id)
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings, None, False, *_cam_inputs(raster_settings), None, points_transform, view_id)


def rasterize_gaussians_raw(means3D, means2D, features_dc, features_rest, opacity_logit, log_scales, rotations_raw,
                            raster_settings, fused_adam=None, points_transform=None, prepared=None, prepare_next=None,
                            next_points_transform=None, densify_stats=None, batch_first_block=None, fused_adam_deferred=False,
                            view_id=0, extras=0):
    """Extension ("next" row f-2): rasterize straight from HTGaussianModel's raw parameters (_xyz, _features_dc,
    _features_rest, _opacity, _scaling, _rotation; /root/reference/scene/gaussian_model_ht.py:74-82) with the
    activations of :49-65,128-133,176-188 fused into the HIP kernels; gradients are w.r.t. the raw tensors.

    fused_adam = a `FusedAdam` whose groups are exactly these six tensors (names xyz, f_dc, f_rest, opacity, scaling,
    rotation as at gaussian_model_ht.py:275-286): backward() then applies that optimizer's step inside the
    per-Gaussian backward kernel (include/gsr.h GsrFusedAdam) and leaves the parameter .grad unset; means2D.grad is
    still produced.  Same result as backward() followed by optimizer.step().

    points_transform = [3,4] / [4,4] tensor M: every mean is replaced by M[:3,:3] p + M[:3,3] inside the kernels -- the
    fused form of `get_xyz` under pose fitting (`self.P[k].retr().act(xyz)`, gaussian_model_ht.py:135-148); its
    gradient comes back through autograd (see pose.py for the SE3 parametrisation).

    prepare_next = the raster settings of the NEXT render of these parameters (with fused_adam): backward() then also runs
    that render's preprocess on the freshly updated parameters (include/gsr.h GsrNextView) and a FIFTH output -- a byte
    buffer, valid once backward() has run -- is returned; hand it to the next call as `prepared=` together with the same
    settings and the (in-place updated) parameter tensors, and that forward skips its preprocess kernel with a bit-identical
    result.  The caller guarantees that nothing else modifies the parameters in between.  next_points_transform = the pose
    transform of that next render when it is not this render's (per-frame poses under refinement).

    densify_stats = (xyz_gradient_accum, denom, max_radii2D), float32 tensors of N elements: backward() then also accumulates
    the per-iteration densification statistics of /root/reference/trainer/ht3dgs_trainer.py:141-147 and
    /root/reference/scene/gaussian_model_ht.py:718-721 into them, inside the per-Gaussian backward kernel (include/gsr.h
    GsrDensifyStats) -- no torch ops on N-sized tensors per step.

    fused_adam_deferred = True (with fused_adam): backward() writes the Adam-updated parameters and moments into the optimizer's
    shadow buffers instead of in place (include/gsr.h GsrFusedAdam::param_out) and `fused_adam.step()` adopts them by swapping
    storages; until then the model is untouched, so surgery or a skipped step between backward() and step() keep the reference's
    meaning (what gsr_autopatch.render_fused uses under the unmodified trainer).

    batch_first_block = [0, b1, ..., N / 128]: the tensors hold B INDEPENDENT models back to back, model k owning the 128-Gaussian
    blocks [b_k, b_k+1) (pad every model to a multiple of 128 with Gaussians that are culled); raster_settings then carries one
    camera per model (viewmatrix / projmatrix [B,4,4], campos [B,3]; points_transform [B,3,4]) and the outputs are [B,3,H,W] /
    [B,1,H,W]: B renders in one launch chain, each bit-identical with rendering its model alone (include/gsr.h GsrBatch; the
    independent single-image fits of stage A, /root/reference/trainer/ht3dgs_trainer.py:697-698).

    view_id = the caller's id of the frame shown (non-zero, the same whenever the same frame is rendered; 0 = none): speed only --
    the forward blend's balanced placement recognises the frame by it instead of by its pose (include/gsr.h GsrForwardArgs::view_id).

    extras (extension binding; not together with prepare_next): bit 0 adds the CLAMPED colour image -- `clamp(color, 0, 1)`, written by
    the blend kernel, differentiable like torch.clamp -- and bit 1 the visibility bytes `radii > 0` (uint8 [N], written by the
    preprocess) to the returned tuple: (color, radii, depth, alpha, clamped, visible) -- what the reference's render wrapper derives
    with a torch launch each (gaussian_model_ht.py:883, :905)."""
    if not E.use_ctypes():
        return _rasterize_ext(means3D, means2D, features_dc, None, opacity_logit, log_scales, rotations_raw, None, raster_settings,
                              features_rest, True, fused_adam, points_transform, prepared, prepare_next, next_points_transform,
                              densify_stats, batch_first_block, fused_adam_deferred, view_id, extras)
    if prepared is not None or extras or prepare_next is not None or densify_stats is not None or batch_first_block is not None or fused_adam_deferred:
        raise RuntimeError("prepared / prepare_next / densify_stats / batch_first_block / fused_adam_deferred are served by the PyTorch extension binding only")
    e = torch.Tensor([])
    return _RasterizeGaussians.apply(means3D, means2D, features_dc, e, opacity_logit, log_scales, rotations_raw, e,
                                     raster_settings, features_rest, True, *_cam_inputs(raster_settings), fused_adam, points_transform, view_id)


class GaussianRasterizer(nn.Module):
    """Callable exactly as at gaussian_model_ht.py:824,871-880 and gaussian_renderer/__init__.py:53,88-96."""

    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """Frustum (near-plane) visibility mask; present in the module's API, unused by the reference."""
        rs = self.raster_settings
        with torch.no_grad():
            dev = positions.device
            if dev.type != "cuda":
                raise RuntimeError("markVisible: tensors must be on a ROCm/HIP device (no CPU fallback)")
            return E.load().mark_visible(positions, rs.viewmatrix.to(dev), rs.projmatrix.to(dev))

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        # (the keyword set is exactly the reference's; extensions such as points_transform live on the functions
        #  rasterize_gaussians / rasterize_gaussians_raw)
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        e = torch.Tensor([])
        return rasterize_gaussians(means3D, means2D, shs if shs is not None else e,
                                   colors_precomp if colors_precomp is not None else e, opacities,
                                   scales if scales is not None else e, rotations if rotations is not None else e,
                                   cov3D_precomp if cov3D_precomp is not None else e, rs)
