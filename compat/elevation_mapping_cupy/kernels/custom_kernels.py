"""Factories of EM/kernels/custom_kernels.py with the reference's signatures, routed to the staged C-ABI calls.

One-to-one: ``dilation_filter_kernel`` (:392-449 -> emap_dilate_planes), ``normal_filter_kernel`` (:452-506 ->
emap_traversability_normals on the given plane), ``polygon_mask_kernel`` (:509-651 -> emap_polygon_mask), ``error_counting_kernel``
(:280-345 -> emap_count + emap_local_drift_sums: the two scalars the drift gate reads).

As a PAIR: ``add_points_kernel`` (:125-277) and ``average_map_kernel`` (:348-389).  The reference communicates between them through
the float accumulator planes ``newmap``; this implementation keeps exact fixed-point accumulators inside the context instead (that
is what makes its results order independent), so ``add_points_kernel``'s callable runs count -> fuse -> commit [-> rays] on a scratch
context, remembers it under ``id(newmap)``, and ``average_map_kernel``'s callable finishes THAT frame (emap_average) and writes the
map back -- the call pattern of ``ElevationMap.update_map_with_kernel`` (EM/elevation_mapping.py:359-375).  ``newmap`` itself is left
untouched: an ``average_map_kernel`` call on accumulator planes that did not come from this package raises.
"""
from __future__ import annotations

import ctypes as ct

import numpy as np

from elevation_mapping_cupy_amd._lib import f32p
from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
from elevation_mapping_cupy_amd.parameter import Parameter

_pending = {}          # id(newmap) -> ElevationMap holding the un-averaged frame


def _scalar(a):
    return float(np.asarray(a, np.float64).reshape(-1)[0])


def _scratch(width, height, resolution=0.04, **fields):
    if int(width) != int(height):
        raise ValueError("square maps only (the reference's maps are square: EM/parameter.py cell_n)")
    p = Parameter()
    p.resolution = float(resolution)
    p.map_length = (int(width) - 2) * float(resolution)
    for k, v in fields.items():
        setattr(p, k, v)
    p.update()
    p.cell_n = int(width)                       # (map_length / resolution can round the other way for odd resolutions)
    p.enable_drift_compensation = True
    return ElevationMap(p)


def _frame_inputs(center_x, center_y, R, t, p):
    t = np.asarray(t, np.float32).reshape(3).copy()
    t[0] -= _scalar(center_x); t[1] -= _scalar(center_y)          # the kernels take the map centre as an argument; the library works centre-relative
    return np.ascontiguousarray(np.asarray(R, np.float32).reshape(3, 3)), t, np.ascontiguousarray(np.asarray(p, np.float32))


def error_counting_kernel(resolution, width, height, sensor_noise_factor, mahalanobis_thresh, outlier_variance, traversability_inlier,
                          min_valid_distance, max_height_range, ramped_height_range_a, ramped_height_range_b, ramped_height_range_c):
    em = _scratch(width, height, resolution, sensor_noise_factor=sensor_noise_factor, mahalanobis_thresh=mahalanobis_thresh,
                  drift_compensation_variance_inlier=outlier_variance, traversability_inlier=traversability_inlier,
                  min_valid_distance=min_valid_distance, max_height_range=max_height_range, ramped_height_range_a=ramped_height_range_a,
                  ramped_height_range_b=ramped_height_range_b, ramped_height_range_c=ramped_height_range_c)
    em.reload_params()

    def kernel(map_, p, center_x, center_y, R, t, newmap, error, error_cnt, size=None):
        R_, t_, pts = _frame_inputs(center_x, center_y, R, t, p)
        em.elevation_map = np.asarray(map_, np.float32)
        em.bind_points(pts)
        em.stage("count", R_, t_)
        s, c = ct.c_double(0.0), ct.c_uint32(0)
        em._chk(em._lib.emap_local_drift_sums(em._ctx, ct.byref(s), ct.byref(c)))
        error[...] = np.asarray(error) + np.float32(s.value)
        error_cnt[...] = np.asarray(error_cnt) + np.float32(c.value)
        em.stage("gate")                                               # re-arms the error slots; no shift (both noises 0)
    return kernel


def add_points_kernel(resolution, width, height, sensor_noise_factor, mahalanobis_thresh, outlier_variance, wall_num_thresh,
                      max_ray_length, cleanup_step, min_valid_distance, max_height_range, cleanup_cos_thresh, ramped_height_range_a,
                      ramped_height_range_b, ramped_height_range_c, enable_edge_shaped=True, enable_visibility_cleanup=True):
    fields = dict(sensor_noise_factor=sensor_noise_factor, mahalanobis_thresh=mahalanobis_thresh, outlier_variance=outlier_variance,
                  wall_num_thresh=wall_num_thresh, max_ray_length=max_ray_length, cleanup_step=cleanup_step,
                  min_valid_distance=min_valid_distance, max_height_range=max_height_range, cleanup_cos_thresh=cleanup_cos_thresh,
                  ramped_height_range_a=ramped_height_range_a, ramped_height_range_b=ramped_height_range_b,
                  ramped_height_range_c=ramped_height_range_c, enable_edge_sharpen=bool(enable_edge_shaped),
                  enable_visibility_cleanup=bool(enable_visibility_cleanup))

    def kernel(center_x, center_y, R, t, norm_map, p, map_, newmap, size=None):
        em = _scratch(width, height, resolution, **fields)
        em.reload_params()
        R_, t_, pts = _frame_inputs(center_x, center_y, R, t, p)
        em.elevation_map = np.asarray(map_, np.float32)
        em.normal_map = np.asarray(norm_map, np.float32)
        em.bind_points(pts)
        em.stage("count", R_, t_); em.stage("gate")                    # the caller has applied the drift shift to `map` already (:353-357)
        em.stage("fuse", R_, t_); em.stage("commit")
        if fields["enable_visibility_cleanup"]:
            em.stage("rays", R_, t_)
        _pending[id(newmap)] = (em, fields["enable_visibility_cleanup"])
    return kernel


def average_map_kernel(width, height, max_variance, initial_variance):
    def kernel(newmap, map_, size=None):
        held = _pending.pop(id(newmap), None)
        if held is None:
            raise NotImplementedError("average_map_kernel: these accumulator planes were not filled by this package's add_points_kernel "
                                      "(the MI355X library keeps its fixed-point accumulators inside the context)")
        em, _rays = held
        em.param.max_variance, em.param.initial_variance = float(max_variance), float(initial_variance)
        em.reload_params()
        em.stage("average")                                            # (ray accumulators are zero when no visibility pass ran)
        map_[...] = em.elevation_map
        em.close()
    return kernel


def dilation_filter_kernel(width, height, dilation_size):
    em = _scratch(width, height)

    def kernel(map_, mask, newmap, newmask, size=None):
        src, msk = np.ascontiguousarray(map_, np.float32), np.ascontiguousarray(mask, np.float32)
        out, omask = np.empty_like(src), np.empty_like(msk)
        em._chk(em._lib.emap_dilate_planes(em._ctx, f32p(src), f32p(msk), int(dilation_size), 1, f32p(out), f32p(omask)))
        newmap[...] = out; newmask[...] = omask
    return kernel


def normal_filter_kernel(width, height, resolution):
    em = _scratch(width, height, resolution)

    def kernel(map_, mask, newmap, size=None):
        # the library's stencil stage starts from the upper-bound plane and dilates it itself: with every cell flagged as having an
        # upper bound the dilation is the identity, and the normal filter sees exactly the plane handed in (is_valid = the mask)
        src = np.ascontiguousarray(map_, np.float32)
        em.set_layer_raw("upper_bound", src)
        em.set_layer_raw("is_upper_bound", np.ones_like(src))
        em.set_layer_raw("is_valid", np.ascontiguousarray(mask, np.float32))
        em.stage("traversability_normals")
        newmap[...] = em.normal_map
    return kernel


def polygon_mask_kernel(width, height, resolution):
    em = _scratch(width, height, resolution)

    def kernel(polygon, center_x, center_y, polygon_n, polygon_bbox, mask, size=None):
        n = int(np.asarray(polygon_n).reshape(-1)[0])
        poly = np.ascontiguousarray(np.asarray(polygon, np.float32).reshape(-1, 2)[:n])
        out = np.empty((int(width), int(height)), np.float32)
        em._chk(em._lib.emap_polygon_mask(em._ctx, f32p(poly), n, ct.c_float(_scalar(center_x)), ct.c_float(_scalar(center_y)), f32p(out)))
        mask[...] = out.reshape(np.asarray(mask).shape)
    return kernel
