"""ctypes binding of oracle/gsr_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see the header of gsr_oracle.c).  PARITY UNPINNED: the rasterizer arithmetic is an un-vendored
third-party submodule of the reference (/root/reference/.gitmodules:4-6); the oracle restates the
published algorithm and is pinned only on the pieces the reference owns (SH, cov3D, matrices) via
tests/golden.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class GsrOracleIn(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("M", C.c_int32), ("D", C.c_int32), ("W", C.c_int32), ("H", C.c_int32),
        ("prefiltered", C.c_int32),
        ("scale_modifier", C.c_float), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
        ("means3D", C.c_void_p), ("scales", C.c_void_p), ("rotations", C.c_void_p),
        ("cov3D_precomp", C.c_void_p), ("opacities", C.c_void_p), ("shs", C.c_void_p),
        ("colors_precomp", C.c_void_p), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p),
        ("campos", C.c_void_p), ("bg", C.c_void_p),
    ]


def build(force=False):
    """Compile the oracle (gcc).  Building the checker is not using it."""
    libs = [os.path.join(_HERE, "libgsr_oracle_f64.so"), os.path.join(_HERE, "libgsr_oracle_f32.so")]
    src = os.path.join(_HERE, "gsr_oracle.c")
    if force or any((not os.path.exists(l)) or os.path.getmtime(l) < os.path.getmtime(src) for l in libs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)
    return libs


_LIBS = {}


def _lib(precision="f64"):
    if precision not in _LIBS:
        build()
        lib = C.CDLL(os.path.join(_HERE, "libgsr_oracle_%s.so" % precision))
        lib.gsr_oracle_forward.restype = C.c_void_p
        lib.gsr_oracle_forward.argtypes = [C.POINTER(GsrOracleIn)] + [C.c_void_p] * 6
        lib.gsr_oracle_backward.restype = None
        lib.gsr_oracle_backward.argtypes = [C.c_void_p] + [C.c_void_p] * 12
        lib.gsr_oracle_free.argtypes = [C.c_void_p]
        lib.gsr_oracle_num_rendered.restype = C.c_int64
        lib.gsr_oracle_num_rendered.argtypes = [C.c_void_p]
        lib.gsr_oracle_pairs.restype = C.c_int64
        lib.gsr_oracle_pairs.argtypes = [C.c_void_p]
        lib.gsr_oracle_get_binning.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.gsr_oracle_get_geom.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        lib.gsr_oracle_run_stages.restype = C.c_int64
        lib.gsr_oracle_run_stages.argtypes = [C.POINTER(GsrOracleIn), C.c_int]
        lib.gsr_oracle_sh_eval.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.gsr_oracle_cov3d.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
        lib.gsr_oracle_set_margins.argtypes = [C.c_double] * 3
        lib.gsr_oracle_resolve_branches.restype = C.c_int64
        lib.gsr_oracle_resolve_branches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 5
        _LIBS[precision] = lib
    return _LIBS[precision]


def _f32(a, shape=None):
    if a is None:
        return None
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleRender:
    """One forward (and optional backward) of the CPU oracle on numpy/torch inputs."""

    def __init__(self, *, means3D, opacities, viewmatrix, projmatrix, campos, bg, image_height,
                 image_width, tanfovx, tanfovy, sh_degree=0, shs=None, colors_precomp=None,
                 scales=None, rotations=None, cov3D_precomp=None, scale_modifier=1.0,
                 prefiltered=False, precision="f64"):
        self.lib = _lib(precision)
        self.means3D = _f32(means3D).reshape(-1, 3)
        N = self.means3D.shape[0]
        self.N, self.W, self.H = N, int(image_width), int(image_height)
        self.opacities = _f32(opacities).reshape(N)
        self.shs = _f32(shs)
        self.M = 0 if self.shs is None else self.shs.reshape(N, -1, 3).shape[1] if N else 16
        self.colors = _f32(colors_precomp)
        self.scales, self.rotations, self.cov3D = _f32(scales), _f32(rotations), _f32(cov3D_precomp)
        assert (self.shs is None) != (self.colors is None)
        assert (self.cov3D is None) != (self.scales is None)
        # matrices are read linearly (storage order of the *contiguous* tensor, scene/cameras.py:76-98)
        self.vm, self.pm = _f32(viewmatrix).reshape(16), _f32(projmatrix).reshape(16)
        self.campos, self.bg = _f32(campos).reshape(3), _f32(bg).reshape(3)
        a = GsrOracleIn()
        a.N, a.M, a.D, a.W, a.H = N, self.M, int(sh_degree), self.W, self.H
        a.prefiltered = int(prefiltered)
        a.scale_modifier, a.tanfovx, a.tanfovy = float(scale_modifier), float(tanfovx), float(tanfovy)
        a.means3D, a.scales, a.rotations = _ptr(self.means3D), _ptr(self.scales), _ptr(self.rotations)
        a.cov3D_precomp, a.opacities, a.shs = _ptr(self.cov3D), _ptr(self.opacities), _ptr(self.shs)
        a.colors_precomp, a.viewmatrix, a.projmatrix = _ptr(self.colors), _ptr(self.vm), _ptr(self.pm)
        a.campos, a.bg = _ptr(self.campos), _ptr(self.bg)
        self.args = a
        self.ctx = None

    def forward(self):
        P = self.W * self.H
        self.color = np.zeros((3, self.H, self.W), np.float32)
        self.depth = np.zeros((1, self.H, self.W), np.float32)
        self.alpha = np.zeros((1, self.H, self.W), np.float32)
        self.radii = np.zeros(self.N, np.int32)
        self.px_ambig = np.zeros((self.H, self.W), np.uint8)
        self.g_ambig = np.zeros(self.N, np.uint8)
        self.ctx = self.lib.gsr_oracle_forward(C.byref(self.args), _ptr(self.color), _ptr(self.depth),
                                               _ptr(self.alpha), _ptr(self.radii), _ptr(self.px_ambig),
                                               _ptr(self.g_ambig))
        self.num_rendered = int(self.lib.gsr_oracle_num_rendered(self.ctx))
        self.pairs = int(self.lib.gsr_oracle_pairs(self.ctx))
        return self.color, self.radii, self.depth, self.alpha

    def resolve_branches(self, got_color, got_alpha, max_events=8):
        """Adopt, per pixel with rounding-edge decisions, the branch closest to the implementation under test (see
        gsr_oracle_resolve_branches).  Replaces self.color / depth / alpha by the adopted branches' outputs; a later
        backward() differentiates the same branches.  Returns dict(events[H,W] uint8, err[H,W] float32, changed)."""
        gc = _f32(got_color, (3, self.H, self.W))
        ga = _f32(got_alpha, (self.H, self.W))
        ev = np.zeros((self.H, self.W), np.uint8)
        err = np.zeros((self.H, self.W), np.float32)
        changed = int(self.lib.gsr_oracle_resolve_branches(self.ctx, _ptr(gc), _ptr(ga), int(max_events), _ptr(ev), _ptr(err),
                                                           _ptr(self.color), _ptr(self.depth), _ptr(self.alpha)))
        return dict(events=ev, err=err, changed=changed)

    def binning(self):
        T = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        ts = np.zeros(T + 1, np.int64)
        lst = np.zeros(max(self.num_rendered, 1), np.uint32)
        self.lib.gsr_oracle_get_binning(self.ctx, _ptr(ts), _ptr(lst))
        return ts, lst[: self.num_rendered]

    def geom(self):
        N = max(self.N, 1)
        xy, depth = np.zeros((N, 2), np.float32), np.zeros(N, np.float32)
        conic, rgb, rect = np.zeros((N, 3), np.float32), np.zeros((N, 3), np.float32), np.zeros((N, 4), np.int32)
        self.lib.gsr_oracle_get_geom(self.ctx, _ptr(xy), _ptr(depth), _ptr(conic), _ptr(rgb), _ptr(rect))
        return dict(xy=xy[: self.N], depth=depth[: self.N], conic=conic[: self.N], rgb=rgb[: self.N],
                    rect=rect[: self.N])

    def backward(self, g_color, g_depth=None, g_alpha=None):
        assert self.ctx is not None
        N, M = self.N, max(self.M, 1)
        gc = _f32(g_color, (3, self.H, self.W)) if g_color is not None else None
        gd = _f32(g_depth, (self.H, self.W)) if g_depth is not None else None
        ga = _f32(g_alpha, (self.H, self.W)) if g_alpha is not None else None
        cam = np.zeros(35)
        out = dict(means3D=np.zeros((N, 3)), means2D=np.zeros((N, 3)), opacities=np.zeros((N, 1)),
                   colors_precomp=np.zeros((N, 3)), shs=np.zeros((N, M, 3)), scales=np.zeros((N, 3)),
                   rotations=np.zeros((N, 4)), cov3D_precomp=np.zeros((N, 6)))
        self.lib.gsr_oracle_backward(self.ctx, _ptr(gc), _ptr(gd), _ptr(ga), _ptr(out["means3D"]),
                                     _ptr(out["means2D"]), _ptr(out["opacities"]),
                                     _ptr(out["colors_precomp"]), _ptr(out["shs"]), _ptr(out["scales"]),
                                     _ptr(out["rotations"]), _ptr(out["cov3D_precomp"]), _ptr(cam))
        out["viewmatrix"], out["projmatrix"], out["campos"] = cam[:16].reshape(4, 4), cam[16:32].reshape(4, 4), cam[32:]
        return out

    def close(self):
        if self.ctx is not None:
            self.lib.gsr_oracle_free(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_stages(render: "OracleRender", with_blend: bool) -> int:
    """cpu_baseline leg: K1-K5 (and K6 when with_blend) on the host cores; returns R."""
    return int(render.lib.gsr_oracle_run_stages(C.byref(render.args), int(with_blend)))


def sh_eval(deg, sh, dirs, precision="f64"):
    """sh [N,16,3] float32, dirs [N,3] float64 -> raw SH colour [N,3] (before +0.5 / clamp)."""
    lib = _lib(precision)
    sh = np.ascontiguousarray(sh, np.float32)
    dirs = np.ascontiguousarray(dirs, np.float64)
    out = np.zeros((sh.shape[0], 3), np.float64)
    for i in range(sh.shape[0]):
        lib.gsr_oracle_sh_eval(int(deg), _ptr(sh[i]), _ptr(dirs[i]), _ptr(out[i]))
    return out


def cov3d(scales, mod, rots, precision="f64"):
    lib = _lib(precision)
    scales = np.ascontiguousarray(scales, np.float32)
    rots = np.ascontiguousarray(rots, np.float32)
    out = np.zeros((scales.shape[0], 6), np.float64)
    for i in range(scales.shape[0]):
        lib.gsr_oracle_cov3d(_ptr(scales[i]), float(mod), _ptr(rots[i]), _ptr(out[i]))
    return out
