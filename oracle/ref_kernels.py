"""TEST INFRASTRUCTURE ONLY.  ctypes front-end for the compiled-reference objects of oracle/build_ref.py.

Call conventions mirror the reference's kernel factories
(reference kernels/custom_kernels.py:125-146, 280-296, 348, 392, 452 and custom_semantic_kernels.py).
All arrays are C-contiguous numpy arrays of the dtype the reference uses (float32 unless noted);
execution is sequential in element order -- one legal interleaving of the racy GPU kernels.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import build_ref

_F = np.float32


def _ptr(a):
    assert a.flags["C_CONTIGUOUS"], "ref kernels need C-contiguous arrays"
    return ctypes.c_void_p(a.ctypes.data)


class RefKernels:
    """One compiled parameter set of the reference kernels."""

    def __init__(self, params, build=True):
        self.params = dict(params)
        path = build_ref.build(self.params) if build else build_ref.so_path(self.params)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = ctypes.CDLL(path)
        self.C = int(params["cell_n"])

    def _call(self, name, arrays, size):
        fn = getattr(self.lib, "ref_" + name)
        fn.restype = None
        fn(*[_ptr(a) for a in arrays], ctypes.c_long(int(size)))

    # --- core kernels -------------------------------------------------------------------------
    def error_counting(self, emap, p, R, t, newmap, error, error_cnt, center=(0.0, 0.0)):
        cx, cy = np.array([center[0]], _F), np.array([center[1]], _F)
        self._call("error_counting", [emap, p, cx, cy, R, t, newmap, error, error_cnt], p.shape[0])

    def add_points(self, R, t, norm_map, p, emap, newmap, center=(0.0, 0.0)):
        cx, cy = np.array([center[0]], _F), np.array([center[1]], _F)
        self._call("add_points", [cx, cy, R, t, norm_map, p, emap, newmap], p.shape[0])

    def average_map(self, newmap, emap):
        self._call("average_map", [newmap, emap], self.C * self.C)

    def dilation_filter(self, plane, mask, out, out_mask, size=None):
        name = "dilation_filter" if size is None else "dilation_filter_%d" % size
        self._call(name, [plane, mask, out, out_mask], self.C * self.C)

    def normal_filter(self, plane, mask, out):
        self._call("normal_filter", [plane, mask, out], self.C * self.C)

    # --- camera path (reference custom_image_kernels.py); scalars are passed by value like the reference does ------
    def image_correspondence(self, emap, x1, y1, z1, P, K, D, image_height, image_width, center, uv, valid):
        fn = self.lib.ref_image_correspondence
        fn.restype = None
        f = ctypes.c_float
        fn(_ptr(emap), f(x1), f(y1), f(z1), _ptr(P), _ptr(K), _ptr(D), f(image_height), f(image_width), _ptr(center),
           _ptr(uv), _ptr(valid), ctypes.c_long(self.C * self.C))

    def image_fuse(self, kind, sem_map, map_idx, image, uv, valid, image_height, image_width, new_sem_map):
        fn = getattr(self.lib, "ref_image_" + kind)
        fn.restype = None
        f = ctypes.c_float
        fn(_ptr(sem_map), f(map_idx), _ptr(image), _ptr(uv), _ptr(valid), f(image_height), f(image_width), _ptr(new_sem_map),
           ctypes.c_long(self.C * self.C))

    def min_filter_sweep(self, elevation, valid, newmap, newmask, size):
        """one in-place sweep of the MinFilter kernel (reference plugins/min_filter.py:29-82), sequential order"""
        self._call("min_filter_%d" % size, [elevation, valid, newmap, newmask], self.C * self.C)

    # --- semantic kernels (reference custom_semantic_kernels.py) ---------------------------------
    def max_filter_sweep(self, elevation, valid, newmap, newmask, size):
        """one sweep of the MaxFilter kernel (reference plugins/max_filter.py:36-93); inputs are copies there (out of place)"""
        self._call("max_filter_%d" % size, [elevation, valid, newmap, newmask], self.C * self.C)

    def sem_sum(self, p, R, t, pcl_chan, map_lay, pcl_channels, smap, newmap, size):
        self._call("sem_sum", [p, R, t, pcl_chan, map_lay, pcl_channels, smap, newmap], size)

    def sem_average(self, newmap, pcl_chan, map_lay, pcl_channels, new_elmap, smap, size):
        self._call("sem_average", [newmap, pcl_chan, map_lay, pcl_channels, new_elmap, smap], size)

    def sem_class_average(self, newmap, pcl_chan, map_lay, pcl_channels, new_elmap, smap, size):
        self._call("sem_class_average", [newmap, pcl_chan, map_lay, pcl_channels, new_elmap, smap], size)

    def sem_add_color(self, p, R, t, pcl_chan, map_lay, pcl_channels, color_map, size):
        self._call("sem_add_color", [p, R, t, pcl_chan, map_lay, pcl_channels, color_map], size)

    def sem_color_average(self, color_map, pcl_chan, map_lay, pcl_channels, smap, size):
        self._call("sem_color_average", [color_map, pcl_chan, map_lay, pcl_channels, smap], size)

    def polygon_mask(self, polygon, center_x, center_y, polygon_bbox, mask):
        """polygon (M, 2) float32; centre scalars; bbox (4,) = [min_x, min_y, max_x, max_y]; mask (C, C) float32 out"""
        cx, cy = np.array([center_x], _F), np.array([center_y], _F)
        n = np.array([polygon.shape[0]], np.int16)
        self._call("polygon_mask", [np.ascontiguousarray(polygon, _F), cx, cy, n, np.ascontiguousarray(polygon_bbox, _F), mask], mask.size)

    # point fusions whose kernels live in the plugin modules (parameter sets with ``bayes_kernels``)
    def alpha(self, p, pcl_chan, map_lay, pcl_channels, newmap, size):
        self._call("alpha", [p, pcl_chan, map_lay, pcl_channels, newmap], size)

    def sum_compact(self, p, R, t, pcl_chan, map_lay, pcl_channels, sum_mean, size):
        self._call("sum_compact", [p, R, t, pcl_chan, map_lay, pcl_channels, sum_mean], size)

    def sem_sum_max(self, p, max_pt, max_id, pcl_chan, map_lay, pcl_channels, newmap, size):
        self._call("sem_sum_max", [p, max_pt, max_id, pcl_chan, map_lay, pcl_channels, newmap], size)

    def bayesian_inference(self, pcl_chan, map_lay, pcl_channels, new_elmap, newmap, sum_mean, smap, size):
        self._call("bayesian_inference", [pcl_chan, map_lay, pcl_channels, new_elmap, newmap, sum_mean, smap], size)


def available(params):
    """True when the object for ``params`` is prebuilt or can be built here."""
    return os.path.exists(build_ref.so_path(params)) or os.path.isdir(build_ref.REF_ROOT)
