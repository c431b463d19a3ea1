"""ctypes binding of ``libemap_hip.so`` (C ABI: ``include/emap_hip.h``).

The product path has NO CPU fallback: if the shared library is missing, cannot be loaded, or no HIP device
is present, the calls below raise -- they never route to NumPy or to the test oracle.
"""
from __future__ import annotations

import ctypes as ct
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EMAP_HIP_LIB") or os.path.join(_PKG, "libemap_hip.so")      # (override: A/B runs of two builds)

_INTS = ["cell_n", "mode", "enable_edge_sharpen", "enable_visibility_cleanup", "enable_drift_compensation",
         "enable_overlap_clearance", "dilation_size", "pad_"]
_DOUBLES = [
    "resolution", "sensor_noise_factor", "mahalanobis_thresh", "outlier_variance",
    "drift_compensation_variance_inlier", "traversability_inlier", "wall_num_thresh", "max_ray_length",
    "cleanup_step", "cleanup_cos_thresh", "min_valid_distance", "max_height_range",
    "ramped_height_range_a", "ramped_height_range_b", "ramped_height_range_c", "max_variance",
    "initial_variance", "time_variance", "time_interval", "min_height_drift_cnt",
    "max_drift", "drift_compensation_alpha", "position_noise_thresh", "orientation_noise_thresh",
    "overlap_clear_range_xy", "overlap_clear_range_z", "ray_step", "reserved_",
]


class EmapParams(ct.Structure):
    """struct emap_params (include/emap_hip.h)."""
    _fields_ = ([(n, ct.c_int32) for n in _INTS] + [(n, ct.c_double) for n in _DOUBLES] +
                [("w1", ct.c_float * 36), ("w2", ct.c_float * 36), ("w3", ct.c_float * 36), ("w_out", ct.c_float * 12)])


class EmapStrip(ct.Structure):
    _fields_ = [("row_begin", ct.c_int32), ("row_count", ct.c_int32), ("halo_rows", ct.c_int32), ("pad_", ct.c_int32)]


class EmapStats(ct.Structure):
    _fields_ = [("err_sum", ct.c_double), ("err_cnt", ct.c_uint32), ("gate_fired", ct.c_int32),
                ("mean_error", ct.c_float), ("additive_mean_error", ct.c_float), ("shift", ct.c_float),
                ("n_points", ct.c_uint32), ("ray_visits", ct.c_uint64)]


class EmapSemSpec(ct.Structure):
    _fields_ = [("n_sum", ct.c_int32), ("sum_chan", ct.c_int32 * 16), ("sum_layer", ct.c_int32 * 16), ("sum_kind", ct.c_int32 * 16),
                ("n_col", ct.c_int32), ("col_chan", ct.c_int32 * 4), ("col_layer", ct.c_int32 * 4), ("alpha", ct.c_double)]


MODE = {"reference_fp16": 0, "fp32": 1}
PLANES = {"elevation": 0, "variance": 1, "is_valid": 2, "traversability": 3, "time": 4, "upper_bound": 5,
          "is_upper_bound": 6, "normal_x": 7, "normal_y": 8, "normal_z": 9, "traversability_input": 10}
STAGES = ["hist", "scan", "scatter", "gate", "fuse", "commit", "rays", "average", "overlap", "post"]

# every symbol include/emap_hip.h declares (checked by tests/test_abi.py without a GPU)
SYMBOLS = [
    "emap_abi_version", "emap_create", "emap_destroy", "emap_set_params", "emap_last_error", "emap_sync", "emap_clear",
    "emap_upload_points", "emap_upload_points_strip", "emap_strip_point_mask", "emap_declare_points_bucketed", "emap_set_points_device", "emap_set_points_device_split", "emap_point_index", "emap_update", "emap_count",
    "emap_set_drift_inputs", "emap_drift_sums_to_device", "emap_set_drift_inputs_device",
    "emap_local_drift_sums", "emap_set_scatter_mode", "emap_fuse", "emap_fuse_average", "emap_commit", "emap_rays", "emap_average", "emap_overlap_clear",
    "emap_dilate", "emap_traversability_normals", "emap_post", "emap_post_part", "emap_update_variance", "emap_update_time", "emap_get_stats",
    "emap_get_layer", "emap_set_layer", "emap_publish_layer", "emap_shift", "emap_strip_logical_begin", "emap_semantic_configure", "emap_semantic_update", "emap_frame_semantics",
    "emap_semantic_get_layer", "emap_semantic_set_layer", "emap_semantic_clear", "emap_semantic_get_alpha", "emap_semantic_set_alpha", "emap_semantic_class_max", "emap_semantic_accumulate", "emap_semantic_finalize", "emap_min_filter", "emap_max_filter", "emap_smooth_filter", "emap_erode", "emap_inpaint_u8", "emap_inpaint_telea_u8", "emap_inpaint_ns_u8", "emap_image_correspondence", "emap_image_get_correspondence",
    "emap_image_fuse", "emap_image_set_tolerance", "emap_image_fuse_arrays", "emap_polygon_mask", "emap_dilate_planes", "emap_halo_bytes", "emap_halo_pack", "emap_halo_unpack", "emap_normal_row_lag", "emap_normal_halo_pack", "emap_normal_halo_unpack", "emap_normal_lag_plan",
    "emap_comm_unique_id", "emap_comm_init", "emap_comm_destroy", "emap_comm_selftest", "emap_comm_count", "emap_comm_wire_bytes", "emap_comm_allreduce_host", "emap_comm_gather_layer", "emap_update_sharded", "emap_set_ray_mode",
    "emap_timer_begin", "emap_timer_end", "emap_enable_stage_timing", "emap_get_stage_times", "emap_last_update_path", "emap_small_frame_aborts",
]

_lib = None
ABI_VERSION = 3      # include/emap_hip.h: EMAP_ABI_VERSION (checked by load(), and across ranks by emap_comm_init)


class EmapError(RuntimeError):
    pass


def load():
    """Load the HIP library or raise (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EmapError(
            "libemap_hip.so is not built (%s). Run `python -m elevation_mapping_cupy_amd.csrc.build` "
            "(needs hipcc, target gfx950). There is no CPU fallback." % LIB_PATH)
    try:
        lib = ct.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the machine
        raise EmapError("cannot load %s: %s (is the ROCm runtime installed?)" % (LIB_PATH, e)) from e
    lib.emap_last_error.restype = ct.c_char_p
    lib.emap_last_error.argtypes = [ct.c_void_p]
    for s in SYMBOLS:
        if s not in ("emap_last_error",):
            try:
                getattr(lib, s).restype = ct.c_int
            except AttributeError:
                if not os.environ.get("EMAP_HIP_LIB"):      # an older build handed in for an A/B run may lack newer entry points
                    raise
    got = int(lib.emap_abi_version())
    if got != ABI_VERSION and not os.environ.get("EMAP_HIP_LIB"):
        raise EmapError("%s speaks ABI version %d, this binding expects %d (include/emap_hip.h: EMAP_ABI_VERSION) -- rebuild with "
                        "`python -m elevation_mapping_cupy_amd.csrc.build`" % (LIB_PATH, got, ABI_VERSION))
    _lib = lib
    return lib


def fill_params(param, cell_n, mode):
    """Turn a ``Parameter`` dataclass into ``struct emap_params``."""
    P = EmapParams()
    P.cell_n = int(cell_n)
    P.mode = MODE[mode]
    for n in ("enable_edge_sharpen", "enable_visibility_cleanup", "enable_drift_compensation", "enable_overlap_clearance"):
        setattr(P, n, int(bool(getattr(param, n))))
    P.dilation_size = int(param.dilation_size)
    for n in _DOUBLES:
        if hasattr(param, n):
            setattr(P, n, float(getattr(param, n)))
    P.ray_step = float(param.resolution) / 2 ** 0.5  # reference custom_kernels.py:268
    for name in ("w1", "w2", "w3", "w_out"):
        arr = np.asarray(getattr(param, name), np.float32).ravel()
        getattr(P, name)[:] = arr.tolist()
    return P


def f32p(a):
    return a.ctypes.data_as(ct.POINTER(ct.c_float))
