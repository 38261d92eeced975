"""ctypes view of the C ABI in include/gsr.h (libgsr_hip.so).  No fallback: if the library is missing the
import of the product path fails loudly."""
import ctypes as C
import os

import torch  # noqa: F401  (must be imported first so libgsr_hip.so binds to torch's libamdhip64)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libgsr_hip.so")

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
GSR_ALLOC_BINNING, GSR_ALLOC_SCRATCH = 0, 1


class GsrForwardArgs(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("M", C.c_int32), ("D", C.c_int32), ("W", C.c_int32), ("H", C.c_int32),
        ("prefiltered", C.c_int32), ("debug", C.c_int32),
        ("scale_modifier", C.c_float), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
        ("means3D", C.c_void_p), ("scales", C.c_void_p), ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p),
        ("opacities", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p),
        ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p), ("bg", C.c_void_p),
        ("out_color", C.c_void_p), ("out_depth", C.c_void_p), ("out_alpha", C.c_void_p), ("radii", C.c_void_p),
        ("geom", C.c_void_p), ("image", C.c_void_p),
        ("alloc", ALLOC_FN), ("alloc_user", C.c_void_p),
        ("shs_rest", C.c_void_p), ("raw_params", C.c_int32),
        ("points_transform", C.c_void_p), ("prepared", C.c_void_p), ("batch", C.c_void_p),
        ("view_id", C.c_int64), ("out_color_clamped", C.c_void_p), ("visible", C.c_void_p),
    ]


class GsrForwardOut(C.Structure):
    _fields_ = [("num_rendered", C.c_int64), ("binning", C.c_void_p), ("binning_bytes", C.c_size_t),
                ("binning_capacity", C.c_int64), ("forward_flags", C.c_int64)]


class GsrBackwardArgs(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("M", C.c_int32), ("D", C.c_int32), ("W", C.c_int32), ("H", C.c_int32),
        ("scale_modifier", C.c_float), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
        ("means3D", C.c_void_p), ("scales", C.c_void_p), ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p),
        ("opacities", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p),
        ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p), ("bg", C.c_void_p),
        ("geom", C.c_void_p), ("image", C.c_void_p), ("binning", C.c_void_p), ("num_rendered", C.c_int64),
        ("grad_color", C.c_void_p), ("grad_depth", C.c_void_p), ("grad_alpha", C.c_void_p),
        ("d_means3D", C.c_void_p), ("d_means2D", C.c_void_p), ("d_opacities", C.c_void_p),
        ("d_colors_precomp", C.c_void_p), ("d_shs", C.c_void_p), ("d_scales", C.c_void_p),
        ("d_rotations", C.c_void_p), ("d_cov3D_precomp", C.c_void_p), ("scratch", C.c_void_p),
        ("shs_rest", C.c_void_p), ("d_shs_rest", C.c_void_p), ("raw_params", C.c_int32),
        ("d_viewmatrix", C.c_void_p), ("d_projmatrix", C.c_void_p), ("d_campos", C.c_void_p),
        ("fused_adam", C.c_void_p),
        ("points_transform", C.c_void_p), ("d_points_transform", C.c_void_p),
        ("binning_capacity", C.c_int64), ("forward_flags", C.c_int64),
        ("next_view", C.c_void_p), ("prepared_out", C.c_void_p), ("densify_stats", C.c_void_p), ("batch", C.c_void_p),
    ]


class GsrFusedAdam(C.Structure):
    _fields_ = [("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("reserved", C.c_int32),
                ("step", C.c_int64), ("lr", C.c_float * 6), ("exp_avg", C.c_void_p * 6), ("exp_avg_sq", C.c_void_p * 6),
                ("step_lag", C.c_int32 * 6), ("param_out", C.c_void_p * 6), ("exp_avg_out", C.c_void_p * 6),
                ("exp_avg_sq_out", C.c_void_p * 6)]


class GsrAdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_uint64), ("lr", C.c_float)]


EXPORTS = [
    "gsr_geom_bytes", "gsr_image_bytes", "gsr_forward_scratch_bytes", "gsr_binning_bytes",
    "gsr_binning_scratch_bytes", "gsr_backward_scratch_bytes", "gsr_forward", "gsr_backward",
    "gsr_mark_visible", "gsr_last_error", "gsr_version", "gsr_set_option", "gsr_sort_pairs_u32",
    "gsr_sort_pairs_u16", "gsr_sort_scratch_bytes", "gsr_image_staged_offset", "gsr_profile_read",
    "gsr_loss_workspace_bytes", "gsr_loss_forward", "gsr_loss_backward", "gsr_adam_step", "gsr_pose_step", "gsr_pose_step_camera",
    "gsr_knn_scratch_bytes", "gsr_knn_mean_dist2", "gsr_get_counter", "gsr_debug_read_binning", "gsr_debug_direct_binning_geometry", "gsr_debug_view_cache_stats", "gsr_prepared_bytes",
    "gsr_prepare_supported", "gsr_prepared_radii_offset", "gsr_stream_copy", "gsr_image_bytes_batched",
    "gsr_masked_max", "gsr_densify_stats_add", "gsr_psnr_scratch_bytes", "gsr_psnr",
    "gsr_loss_workspace_bytes_batched", "gsr_loss_forward_batched", "gsr_loss_backward_batched", "gsr_loss_forward_terms", "gsr_pose_grad", "gsr_struct_bytes", "gsr_debug_list_cut_stats",
]

_lib = None


def load():
    """Load libgsr_hip.so (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built (run `python __graft_entry__.py` or "
            "`python 3dgs_hierarchical_training_amd/build.py`).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.gsr_prepare_supported.restype = C.c_int
    lib.gsr_prepare_supported.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    for fn in ["gsr_geom_bytes", "gsr_forward_scratch_bytes", "gsr_backward_scratch_bytes", "gsr_prepared_bytes", "gsr_prepared_radii_offset"]:
        getattr(lib, fn).restype = C.c_size_t
        getattr(lib, fn).argtypes = [C.c_int32]
    lib.gsr_image_bytes.restype = C.c_size_t
    lib.gsr_image_bytes.argtypes = [C.c_int32, C.c_int32]
    lib.gsr_image_bytes_batched.restype = C.c_size_t
    lib.gsr_image_bytes_batched.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.gsr_image_staged_offset.restype = C.c_size_t
    lib.gsr_image_staged_offset.argtypes = [C.c_int32, C.c_int32]
    lib.gsr_profile_read.restype = C.c_int
    lib.gsr_profile_read.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.gsr_loss_workspace_bytes.restype = C.c_size_t
    lib.gsr_loss_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.gsr_loss_forward.restype = C.c_int
    lib.gsr_loss_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32,
                                     C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gsr_loss_backward.restype = C.c_int
    lib.gsr_loss_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gsr_knn_scratch_bytes.restype = C.c_size_t
    lib.gsr_knn_scratch_bytes.argtypes = [C.c_int32]
    lib.gsr_knn_mean_dist2.restype = C.c_int
    lib.gsr_knn_mean_dist2.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.gsr_adam_step.restype = C.c_int
    lib.gsr_adam_step.argtypes = [C.POINTER(GsrAdamTensor), C.c_int32, C.c_float, C.c_float, C.c_float, C.c_int64, C.c_void_p]
    lib.gsr_pose_step.restype = C.c_int
    lib.gsr_pose_step.argtypes = [C.c_void_p] * 6 + [C.c_float] * 4 + [C.c_int64, C.c_void_p]
    lib.gsr_pose_step_camera.restype = C.c_int
    lib.gsr_pose_step_camera.argtypes = [C.c_void_p] * 11 + [C.c_float] * 4 + [C.c_int64, C.c_void_p]
    lib.gsr_binning_bytes.restype = C.c_size_t
    lib.gsr_binning_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32]
    lib.gsr_binning_scratch_bytes.restype = C.c_size_t
    lib.gsr_binning_scratch_bytes.argtypes = [C.c_int64]
    lib.gsr_sort_scratch_bytes.restype = C.c_size_t
    lib.gsr_sort_scratch_bytes.argtypes = [C.c_uint32]
    lib.gsr_forward.restype = C.c_int
    lib.gsr_forward.argtypes = [C.POINTER(GsrForwardArgs), C.POINTER(GsrForwardOut), C.c_void_p]
    lib.gsr_backward.restype = C.c_int
    lib.gsr_backward.argtypes = [C.POINTER(GsrBackwardArgs), C.c_void_p]
    lib.gsr_mark_visible.restype = C.c_int
    lib.gsr_mark_visible.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gsr_last_error.restype = C.c_char_p
    lib.gsr_version.restype = C.c_int
    lib.gsr_struct_bytes.restype = C.c_size_t
    lib.gsr_struct_bytes.argtypes = [C.c_int32]
    # the ctypes mirrors above must be the structs the library was compiled with (a shorter struct makes the library
    # write through garbage pointers)
    for which, cls in ((0, GsrForwardArgs), (1, GsrBackwardArgs), (2, GsrForwardOut)):
        if lib.gsr_struct_bytes(which) != C.sizeof(cls):
            raise RuntimeError(f"libgsr_hip.so was built from another include/gsr.h: {cls.__name__} is {lib.gsr_struct_bytes(which)} bytes "
                               f"in the library, {C.sizeof(cls)} in _lib.py")
    lib.gsr_pose_grad.restype = C.c_int
    lib.gsr_pose_grad.argtypes = [C.c_void_p] * 5
    lib.gsr_set_option.restype = C.c_int
    lib.gsr_set_option.argtypes = [C.c_char_p, C.c_int]
    lib.gsr_stream_copy.restype = C.c_int
    lib.gsr_stream_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    lib.gsr_get_counter.restype = C.c_int64
    lib.gsr_get_counter.argtypes = [C.c_char_p]
    lib.gsr_debug_direct_binning_geometry.restype = C.c_int
    lib.gsr_debug_direct_binning_geometry.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_int64)]
    lib.gsr_debug_view_cache_stats.restype = C.c_int
    lib.gsr_debug_view_cache_stats.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_int64)]
    lib.gsr_debug_list_cut_stats.restype = C.c_int
    lib.gsr_debug_list_cut_stats.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_int64)]
    lib.gsr_debug_read_binning.restype = C.c_int
    lib.gsr_debug_read_binning.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    for fn in ["gsr_sort_pairs_u32", "gsr_sort_pairs_u16"]:
        getattr(lib, fn).restype = C.c_int
        getattr(lib, fn).argtypes = [C.c_void_p] * 4 + [C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                                        C.POINTER(C.c_int), C.c_void_p]
    # GSR_OPTS="name=value,name=value": library options applied once, at load -- how A/B tools and test runs put a non-default route under
    # every caller of the process (tools/ab_*.sh; `GSR_OPTS=tile_sort=2 pytest tests -m gpu`).  An option the library refuses is an error.
    for kv in os.environ.get("GSR_OPTS", "").split(","):
        if "=" in kv:
            k, v = kv.split("=", 1)
            if lib.gsr_set_option(k.strip().encode(), int(v)) != 0:
                raise RuntimeError(f"GSR_OPTS: gsr_set_option({k.strip()!r}, {v}) refused")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {load().gsr_last_error().decode()}")
