"""Factories of EM/kernels/custom_semantic_kernels.py with the reference's signatures, routed to the staged C-ABI calls
``emap_semantic_accumulate`` / ``emap_semantic_finalize`` (include/emap_hip.h), which run the same raw-array elementwise kernels on
the MI355X: the points carry (cell index, valid, inside) in their first three columns, exactly as the reference's callers prepare
them (EM/fusion/pointcloud_*.py, EM/tests/test_semantic_kernels.py:25-307).  Every factory returns ``k(*arrays, size=n)`` on NumPy
arrays; output arrays are updated in place like a CuPy ``raw`` out-parameter.

The per-frame path does not go through these: ``SemanticMap.update_layers_pointcloud`` fuses accumulate + finalise per map tile in
LDS (``emap_semantic_update``).  Both are pinned against the reference's own kernels (tests/golden/semantic_toy.npz,
semantic_yaml66.npz, bayes_yaml66.npz).
"""
from __future__ import annotations

import ctypes as ct

import numpy as np

from elevation_mapping_cupy_amd import _lib
from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
from elevation_mapping_cupy_amd.parameter import Parameter

ACC = {"sum": 0, "sum_compact": 1, "sum_max": 2, "alpha": 3, "add_color": 4}
FIN = {"average": 0, "class_average": 1, "bayesian_inference": 2, "color_average": 3}
_scratch_ctx = None


def _ctx():
    """one small context for device / stream ownership (the kernels work on the caller's arrays, not on a map)"""
    global _scratch_ctx
    if _scratch_ctx is None:
        p = Parameter()
        p.update()
        p.cell_n = 8
        _scratch_ctx = ElevationMap(p)
    return _scratch_ctx


def _i32(a):
    return np.ascontiguousarray(np.asarray(a).reshape(-1), np.int32)


def _ip(a):
    return a.ctypes.data_as(ct.POINTER(ct.c_int32))


def _inplace(arr, dtype):
    """the caller's array as a C-contiguous buffer of ``dtype``; returns (buffer, write_back) -- write_back copies into ``arr`` when a
    conversion was needed"""
    a = np.asarray(arr)
    if a.dtype == dtype and a.flags["C_CONTIGUOUS"]:
        return a, lambda: None
    b = np.ascontiguousarray(a, dtype)

    def back():
        arr[...] = b.reshape(a.shape)
    return b, back


def _chk(rc):
    em = _ctx()
    if rc != 0:
        raise _lib.EmapError("libemap_hip call failed (%d): %s" % (rc, em._lib.emap_last_error(em._ctx).decode()))


def _accumulate(op, width, height):
    cells = int(width) * int(height)

    def kernel(p, *rest, size):
        em = _ctx()
        pts = np.ascontiguousarray(p, np.float32)
        if op == "sum_max":                      # (p, max_pt, max_id, pcl_chan, map_lay, pcl_channels, newmap)
            max_pt, max_id, pcl_chan, map_lay, pcl_channels, newmap = rest
            n_max = int(np.asarray(pcl_channels).reshape(-1)[2])
            mp, mi = np.ascontiguousarray(max_pt, np.float32), _i32(max_id)
        elif op == "alpha":                      # (p, pcl_chan, map_lay, pcl_channels, newmap)
            pcl_chan, map_lay, pcl_channels, newmap = rest
            n_max, mp, mi = 0, None, None
        elif op == "sum":                        # (p, R, t, pcl_chan, map_lay, pcl_channels, map, newmap): R, t and map are not read by the reference kernel
            _R, _t, pcl_chan, map_lay, pcl_channels, _map, newmap = rest
            n_max, mp, mi = 0, None, None
        else:                                    # sum_compact / add_color: (p, R, t, pcl_chan, map_lay, pcl_channels, newmap | color_map)
            _R, _t, pcl_chan, map_lay, pcl_channels, newmap = rest
            n_max, mp, mi = 0, None, None
        pc, ml = _i32(pcl_chan), _i32(map_lay)
        ch = np.asarray(pcl_channels).reshape(-1)
        stride, n_ch = int(ch[0]), int(ch[1])
        buf, back = _inplace(newmap, np.uint32 if op == "add_color" else np.float32)
        _chk(em._lib.emap_semantic_accumulate(
            em._ctx, ACC[op], _lib.f32p(pts), ct.c_int64(pts.size // stride), stride, _ip(pc), _ip(ml), n_ch, ct.c_int64(int(size)), ct.c_int64(cells),
            buf.ctypes.data_as(ct.c_void_p), int(buf.size // cells), _lib.f32p(mp) if mp is not None else None, _ip(mi) if mi is not None else None, n_max))
        back()
    kernel.__name__ = op + "_kernel"
    return kernel


def _finalize(op, width, height, alpha=0.0):
    cells = int(width) * int(height)

    def kernel(*args, size):
        em = _ctx()
        sum_mean = None
        if op == "bayesian_inference":           # (pcl_chan, map_lay, pcl_channels, new_elmap, newmap, sum_mean, map)
            pcl_chan, map_lay, pcl_channels, new_elmap, newmap, sum_mean, smap = args
        elif op == "color_average":              # (color_map, pcl_chan, map_lay, pcl_channels, map)
            newmap, pcl_chan, map_lay, pcl_channels, smap = args
            new_elmap = None
        else:                                    # average / class_average: (newmap, pcl_chan, map_lay, pcl_channels, new_elmap, map)
            newmap, pcl_chan, map_lay, pcl_channels, new_elmap, smap = args
        ml = _i32(map_lay)
        n_ch = int(np.asarray(pcl_channels).reshape(-1)[1])
        nm, nm_back = _inplace(newmap, np.uint32 if op == "color_average" else np.float32)
        mp, mp_back = _inplace(smap, np.float32)
        el = np.ascontiguousarray(np.asarray(new_elmap, np.float32)[:3]) if new_elmap is not None else None
        sm = np.ascontiguousarray(sum_mean, np.float32) if sum_mean is not None else None
        _chk(em._lib.emap_semantic_finalize(
            em._ctx, FIN[op], nm.ctypes.data_as(ct.c_void_p), int(nm.size // cells), _ip(ml), n_ch, ct.c_int64(int(size)), ct.c_int64(cells),
            _lib.f32p(el) if el is not None else None, _lib.f32p(sm) if sm is not None else None, int(sm.size // cells) if sm is not None else 0,
            _lib.f32p(mp), int(mp.size // cells), ct.c_double(float(alpha))))
        mp_back()
        if op == "bayesian_inference":
            nm_back()
    kernel.__name__ = op + "_kernel"
    return kernel


def sum_kernel(resolution, width, height):
    return _accumulate("sum", width, height)


def sum_compact_kernel(resolution, width, height):
    return _accumulate("sum_compact", width, height)


def sum_max_kernel(resolution, width, height):
    return _accumulate("sum_max", width, height)


def alpha_kernel(resolution, width, height):
    return _accumulate("alpha", width, height)


def add_color_kernel(width, height):
    return _accumulate("add_color", width, height)


def average_kernel(width, height):
    return _finalize("average", width, height)


def class_average_kernel(width, height, alpha):
    return _finalize("class_average", width, height, alpha)


def bayesian_inference_kernel(width, height):
    return _finalize("bayesian_inference", width, height)


def color_average_kernel(width, height):
    return _finalize("color_average", width, height)
