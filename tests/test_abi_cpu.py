"""CPU: the C-ABI library loads and exports every symbol include/gsr.h declares; the product path refuses to
run without a GPU or without the built extension (no fallback)."""
import ctypes as C
import importlib
import os
import re
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_and_exports():
    b = importlib.import_module("3dgs_hierarchical_training_amd.build")
    lib_path = b.build()
    assert os.path.exists(lib_path)
    L = importlib.import_module("3dgs_hierarchical_training_amd._lib")
    lib = L.load()
    hdr = open(os.path.join(REPO, "include", "gsr.h")).read()
    declared = set(re.findall(r"\b(gsr_[a-z0-9_]+)\s*\(", hdr)) - {"gsr_alloc_fn"}
    assert declared, "no declarations parsed"
    for sym in declared:
        assert hasattr(lib, sym), f"libgsr_hip.so does not export {sym}"
    assert set(L.EXPORTS) <= declared | {"gsr_forward_scratch_bytes"}
    assert lib.gsr_version() >= 100
    assert lib.gsr_geom_bytes(10) >= 480 and lib.gsr_image_bytes(64, 64) >= 64 * 64 * 28


def test_struct_layout_matches_header():
    """ctypes mirrors of the argument structs have the field order of include/gsr.h."""
    L = importlib.import_module("3dgs_hierarchical_training_amd._lib")
    hdr = open(os.path.join(REPO, "include", "gsr.h")).read()
    for name, cls in [("GsrForwardArgs", L.GsrForwardArgs), ("GsrBackwardArgs", L.GsrBackwardArgs), ("GsrForwardOut", L.GsrForwardOut),
                      ("GsrFusedAdam", L.GsrFusedAdam)]:
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for stmt in body.split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            for part in stmt.split(","):
                fields.append(re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[\d+\])?$", part.strip())[0])
        assert fields == [f[0] for f in cls._fields_], name


def test_cpu_tensors_are_refused():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    rs = GaussianRasterizationSettings(image_height=16, image_width=16, tanfovx=0.5, tanfovy=0.5, bg=torch.zeros(3),
                                       scale_modifier=1.0, viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0,
                                       campos=torch.zeros(3), prefiltered=False, debug=False)
    r = GaussianRasterizer(rs)
    x = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        r(means3D=x, means2D=x, shs=None, colors_precomp=torch.zeros(4, 3), opacities=torch.ones(4, 1),
          scales=torch.ones(4, 3), rotations=torch.ones(4, 4), cov3D_precomp=None)
    with pytest.raises(Exception, match="excatly one"):
        r(means3D=x, means2D=x, shs=None, colors_precomp=None, opacities=torch.ones(4, 1), scales=torch.ones(4, 3),
          rotations=torch.ones(4, 4), cov3D_precomp=None)
    with pytest.raises(Exception, match="exactly one"):
        r(means3D=x, means2D=x, shs=None, colors_precomp=x, opacities=torch.ones(4, 1), scales=None, rotations=None,
          cov3D_precomp=None)


def test_missing_library_fails_loudly(monkeypatch):
    L = importlib.import_module("3dgs_hierarchical_training_amd._lib")
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libgsr_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.load()


def test_product_path_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under the package or the drop-in alias may reference it."""
    pkg = os.path.join(REPO, "3dgs_hierarchical_training_amd")
    for root in (pkg, os.path.join(REPO, "diff_gaussian_rasterization"), os.path.join(REPO, "simple_knn")):
        for dp, _, fns in os.walk(root):
            for fn in fns:
                if fn.endswith((".py", ".hip", ".h")):
                    txt = open(os.path.join(dp, fn)).read()
                    assert "import oracle" not in txt and "from oracle" not in txt and "hostemu" not in txt.replace("tests/hostemu", ""), fn


def test_direct_binning_geometry_invariants():
    """Host logic of the direct binning (csrc/gsr_kernels.hip direct_bin_geometry): chunks of whole 64-Gaussian steps that cover N,
    u16 counters that cannot overflow (a chunk, and a group of chunks, below 65 536 Gaussians), at most 64 groups, frames above
    4 096 tiles left to the sort route.  No GPU needed."""
    L = importlib.import_module("3dgs_hierarchical_training_amd._lib")
    lib = L.load()
    out = (C.c_int64 * 7)()
    for N in (1, 63, 64, 65, 1000, 20_000, 130_000, 1_000_000, 4_000_000, 30_000_000):
        for T in (1, 64, 256, 2170, 4096):
            assert lib.gsr_debug_direct_binning_geometry(N, T, out) == 0
            ok, S, NC, G, Cg, Tp, nbytes = (int(v) for v in out)
            if not ok:
                assert N > 4_000_000      # (only a model too large for 16-bit chunk counters is refused at these frame sizes)
                continue
            assert S % 64 == 0 and S >= 128 and S <= 65472
            assert NC * S >= N > (NC - 1) * S
            assert G * Cg >= NC > (G - 1) * Cg and 1 <= G <= 64
            assert Cg * S <= 65535
            assert Tp % 64 == 0 and T <= Tp < T + 64
            assert nbytes >= NC * Tp * 2 + G * Tp * 4 + (T + 1) * 4
    for N, T in ((0, 100), (1000, 0), (1000, 4097), (1000, 8160)):
        assert lib.gsr_debug_direct_binning_geometry(N, T, out) == 0 and out[0] == 0



def test_tile_sort_option_values():
    """`tile_sort` takes 0 / 1 / 2 (off / where the lists are short / wherever the direct binning runs) and nothing else; its threshold
    is a non-negative count.  No GPU needed: options are host state."""
    L = importlib.import_module("3dgs_hierarchical_training_amd._lib")
    lib = L.load()
    try:
        for v in (0, 2, 1):
            assert lib.gsr_set_option(b"tile_sort", v) == 0
        assert lib.gsr_set_option(b"tile_sort", 3) != 0 and lib.gsr_set_option(b"tile_sort", -1) != 0
        assert lib.gsr_set_option(b"tile_sort_max_avg", 0) == 0 and lib.gsr_set_option(b"tile_sort_max_avg", 800) == 0
        assert lib.gsr_set_option(b"tile_sort_max_avg", -5) != 0
    finally:
        lib.gsr_set_option(b"tile_sort", 1)
        lib.gsr_set_option(b"tile_sort_max_avg", 700)


def test_options_from_the_environment_at_load():
    """`GSR_OPTS=name=value,...` is applied by _lib.load() (a whole test run or bench under a non-default route of the library); a name or
    value the library refuses is an error at load, not a silently ignored word.  Own processes: the options are process state."""
    import subprocess
    code = ("import importlib, sys; sys.path.insert(0, '.'); L = importlib.import_module('3dgs_hierarchical_training_amd._lib'); "
            "lib = L.load(); print('loaded', lib.gsr_set_option(b'tile_sort', 1))")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ok = subprocess.run([sys.executable, "-c", code], cwd=root, env={**os.environ, "GSR_OPTS": "tile_sort=2, blend_bwd_ppt=1"}, capture_output=True, text=True)
    assert ok.returncode == 0 and "loaded 0" in ok.stdout, ok.stderr[-400:]
    for bad in ("no_such_option=1", "tile_sort=7"):
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env={**os.environ, "GSR_OPTS": bad}, capture_output=True, text=True)
        assert r.returncode != 0 and "GSR_OPTS" in r.stderr, (bad, r.stdout, r.stderr[-400:])
