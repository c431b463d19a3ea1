"""TEST INFRASTRUCTURE ONLY.  ctypes front-end + numpy state holder for ``oracle/emap_oracle.c``.

``OracleMap`` mirrors the slice of the reference's ``ElevationMap`` that lies on the hot path
(reference elevation_mapping.py:316-426) with planar ``(7, C, C)`` float32 state, so parity tests can put
it next to the HIP ``ElevationMap`` and compare plane by plane.
"""
from __future__ import annotations

import ctypes as ct
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(HERE, "_build", "libemap_oracle.so")
_SRC = os.path.join(HERE, "emap_oracle.c")

_DOUBLES = [
    "resolution", "sensor_noise_factor", "mahalanobis_thresh", "outlier_variance",
    "drift_compensation_variance_inlier", "traversability_inlier", "wall_num_thresh", "max_ray_length",
    "cleanup_step", "cleanup_cos_thresh", "min_valid_distance", "max_height_range",
    "ramped_height_range_a", "ramped_height_range_b", "ramped_height_range_c", "max_variance",
    "initial_variance", "time_variance", "time_interval", "min_height_drift_cnt",
    "max_drift", "drift_compensation_alpha", "position_noise_thresh", "orientation_noise_thresh",
    "overlap_clear_range_xy", "overlap_clear_range_z", "ray_step", "reserved_",
]
_INTS = ["cell_n", "mode", "enable_edge_sharpen", "enable_visibility_cleanup", "enable_drift_compensation",
         "enable_overlap_clearance", "dilation_size", "pad_"]


class EoParams(ct.Structure):
    _fields_ = ([(n, ct.c_int32) for n in _INTS] + [(n, ct.c_double) for n in _DOUBLES] +
                [("w1", ct.c_float * 36), ("w2", ct.c_float * 36), ("w3", ct.c_float * 36), ("w_out", ct.c_float * 12)])


class EoStats(ct.Structure):
    _fields_ = [("err_sum", ct.c_double), ("err_cnt", ct.c_uint32), ("gate_fired", ct.c_int32),
                ("mean_error", ct.c_float), ("shift", ct.c_float), ("ray_visits", ct.c_uint64)]


# Parameter defaults of the reference dataclass (parameter.py:137-216) for the fields the path uses.
DEFAULTS = dict(
    resolution=0.04, map_length=8.0, sensor_noise_factor=0.05, mahalanobis_thresh=2.0, outlier_variance=0.01,
    drift_compensation_variance_inlier=0.1, traversability_inlier=0.1, wall_num_thresh=100, max_ray_length=2.0,
    cleanup_step=0.01, cleanup_cos_thresh=0.5, min_valid_distance=0.3, max_height_range=1.0,
    ramped_height_range_a=0.3, ramped_height_range_b=1.0, ramped_height_range_c=0.2, max_variance=1.0,
    initial_variance=10.0, time_variance=0.01, time_interval=0.1, min_height_drift_cnt=100, max_drift=0.1,
    drift_compensation_alpha=1.0, position_noise_thresh=0.1, orientation_noise_thresh=0.1,
    overlap_clear_range_xy=4.0, overlap_clear_range_z=2.0, enable_edge_sharpen=True,
    enable_visibility_cleanup=True, enable_drift_compensation=True, enable_overlap_clearance=True,
    dilation_size=2, average_weight=0.5,
)
# elevation_mapping_cupy/config/core/core_param.yaml
YAML = dict(
    DEFAULTS, traversability_inlier=0.9, wall_num_thresh=20, max_ray_length=10.0, cleanup_step=0.1,
    cleanup_cos_thresh=0.1, min_valid_distance=0.5, max_variance=100.0, initial_variance=1000.0,
    time_variance=0.0001, drift_compensation_alpha=0.1, position_noise_thresh=0.01,
    orientation_noise_thresh=0.01, dilation_size=3,
)


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        # -mfma: the explicit fmaf() calls of the traversability filter become one instruction (without it glibc's fmaf is called:
        # same result, slower); contraction of ordinary a * b + c stays off
        try:
            hw_fma = " fma " in open("/proc/cpuinfo").read().replace("\n", " ")
        except OSError:
            hw_fma = False
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-fno-fast-math", "-std=c11", "-shared", "-fPIC"] +
                              (["-mfma"] if hw_fma else []) + [_SRC, "-o", _SO, "-lm"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ct.CDLL(build())
        _lib.eo_gate.restype = ct.c_float
        _lib.eo_f32_to_f16.restype = ct.c_uint16
        _lib.eo_f32_to_f16.argtypes = [ct.c_float]
        _lib.eo_f16_to_f32.restype = ct.c_float
        _lib.eo_f16_to_f32.argtypes = [ct.c_uint16]
    return _lib


def make_params(cfg, cell_n=None, mode="reference_fp16", weights=None):
    """cfg: dict with the reference's Parameter field names. cell_n defaults to round(map_length/res)+2
    (parameter.py:282-289)."""
    full = dict(DEFAULTS)
    full.update(cfg)
    P = EoParams()
    P.cell_n = int(cell_n if cell_n is not None else full.get("cell_n") or
                   int(round(full["map_length"] / full["resolution"])) + 2)
    P.mode = {"reference_fp16": 0, "fp32": 1}[mode]
    for n in _INTS[2:7]:
        setattr(P, n, int(full[n]))
    for n in _DOUBLES:
        if n in full:
            setattr(P, n, float(full[n]))
    P.ray_step = float(full["resolution"]) / 2 ** 0.5  # custom_kernels.py:268
    if weights is not None:
        for name, key in (("w1", "w1"), ("w2", "w2"), ("w3", "w3"), ("w_out", "w_out")):
            arr = np.asarray(weights[key], np.float32).ravel()
            getattr(P, name)[:] = arr.tolist()
    return P


def _p(a):
    return ct.c_void_p(a.ctypes.data) if a is not None else ct.c_void_p(0)


def _pts(points):
    pts = np.ascontiguousarray(points, np.float32)
    assert pts.ndim == 2 and pts.shape[1] >= 3
    return pts


class OracleMap:
    """Numpy-state restatement of the hot-path part of the reference ``ElevationMap``."""

    def __init__(self, P: EoParams):
        self.P, self.C = P, P.cell_n
        C = self.C
        self.elevation_map = np.zeros((7, C, C), np.float32)
        self.elevation_map[1] += np.float32(P.initial_variance)
        self.elevation_map[3] += 1.0
        self.normal_map = np.zeros((3, C, C), np.float32)
        self.traversability_input = np.zeros((C, C), np.float32)
        self.mean_error = 0.0
        self.additive_mean_error = np.float32(0.0)
        self.last = {}

    # ---- stage-by-stage (each stage returns/keeps its accumulators in self.last) -----------------
    def point_index(self, points, R, t):
        pts = _pts(points)
        n = pts.shape[0]
        idx, valid, inside = np.zeros(n, np.int32), np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        R = np.ascontiguousarray(R, np.float32).ravel(); t = np.ascontiguousarray(t, np.float32)
        lib().eo_point_index(ct.byref(self.P), _p(pts), ct.c_long(n), ct.c_long(pts.shape[1]), _p(R), _p(t),
                             _p(idx), _p(valid), _p(inside))
        return idx, valid, inside

    def count(self, points, R, t):
        pts, C = _pts(points), self.C
        R = np.ascontiguousarray(R, np.float32).ravel(); t = np.ascontiguousarray(t, np.float32)
        n_pts, n_inl = np.zeros((C, C), np.uint32), np.zeros((C, C), np.uint32)
        err, cnt = ct.c_double(0.0), ct.c_uint32(0)
        lib().eo_count(ct.byref(self.P), _p(self.elevation_map), _p(pts), ct.c_long(pts.shape[0]),
                       ct.c_long(pts.shape[1]), _p(R), _p(t), _p(n_pts), _p(n_inl), ct.byref(err), ct.byref(cnt))
        self.last.update(n_pts=n_pts, n_inl=n_inl, err_sum=err.value, err_cnt=cnt.value)
        return n_pts, n_inl, err.value, cnt.value

    def gate(self, position_noise, orientation_noise):
        mean, fired = ct.c_float(0.0), ct.c_int(0)
        shift = lib().eo_gate(ct.byref(self.P), ct.c_double(self.last["err_sum"]), ct.c_uint32(self.last["err_cnt"]),
                              ct.c_double(position_noise), ct.c_double(orientation_noise), ct.byref(mean), ct.byref(fired))
        if fired.value:
            self.mean_error = np.float32(mean.value)
            self.additive_mean_error = np.float32(self.additive_mean_error + np.float32(mean.value))
        if shift != 0.0:
            self.elevation_map[0] += np.float32(shift)
        self.last.update(shift=shift, gate_fired=bool(fired.value))
        return shift

    def fuse(self, points, R, t):
        pts, C = _pts(points), self.C
        R = np.ascontiguousarray(R, np.float32).ravel(); t = np.ascontiguousarray(t, np.float32)
        sum_h, sum_v = np.zeros((C, C), np.int64), np.zeros((C, C), np.int64)      # fixed point: Q31.32, Q23.40 (emap_oracle.c)
        cnt, n_out, latest = np.zeros((C, C), np.uint32), np.zeros((C, C), np.uint32), np.zeros((C, C), np.float32)
        lib().eo_fuse(ct.byref(self.P), _p(self.elevation_map), _p(pts), ct.c_long(pts.shape[0]), ct.c_long(pts.shape[1]),
                      _p(R), _p(t), _p(self.last["n_pts"]), _p(sum_h), _p(sum_v), _p(cnt), _p(n_out), _p(latest))
        self.last.update(sum_h=sum_h, sum_v=sum_v, cnt=cnt, n_out=n_out, latest=latest)

    def commit(self):
        L = self.last
        lib().eo_commit(ct.byref(self.P), _p(self.elevation_map), _p(L["cnt"]), _p(L["n_out"]), _p(L["latest"]))

    def rays(self, points, R, t):
        pts, C = _pts(points), self.C
        R = np.ascontiguousarray(R, np.float32).ravel(); t = np.ascontiguousarray(t, np.float32)
        dec, hits = np.zeros((C, C), np.int64), np.zeros((C, C), np.uint32)          # dec: Q23.40
        upper = np.full((C, C), np.inf, np.float32)
        visits = ct.c_uint64(0)
        lib().eo_rays(ct.byref(self.P), _p(self.elevation_map), _p(self.normal_map), _p(self.last["n_inl"]), _p(pts),
                      ct.c_long(pts.shape[0]), ct.c_long(pts.shape[1]), _p(R), _p(t), _p(dec), _p(hits), _p(upper),
                      ct.byref(visits))
        self.last.update(ray_dec=dec, ray_hits=hits, ray_upper=upper, ray_visits=visits.value)

    def average(self):
        L = self.last
        lib().eo_average(ct.byref(self.P), _p(self.elevation_map), _p(L["sum_h"]), _p(L["sum_v"]), _p(L["cnt"]),
                         _p(L.get("ray_dec")), _p(L.get("ray_hits")), _p(L.get("ray_upper")))
        for k in ("ray_dec", "ray_hits", "ray_upper"):
            L.pop(k, None)

    def overlap_clear(self, tz):
        lib().eo_overlap_clear(ct.byref(self.P), _p(self.elevation_map), ct.c_float(tz))

    def dilate(self):
        m = self.elevation_map
        mask = (m[2] + m[6]).astype(np.float32)
        self.traversability_input[...] = 0
        lib().eo_dilate(ct.c_int(self.C), ct.c_int(self.P.dilation_size), _p(np.ascontiguousarray(m[5])), _p(mask),
                        _p(self.traversability_input), ct.c_void_p(0))

    def traversability(self):
        lib().eo_traversability(ct.byref(self.P), _p(self.traversability_input), _p(self.elevation_map[3]))

    def normals(self):
        lib().eo_normals(ct.byref(self.P), _p(self.traversability_input), _p(self.elevation_map[2]), _p(self.normal_map))

    # ---- semantic point fusion (reference semantic_map.py:223-259 + fusion/pointcloud_*.py) ----------
    def semantic_update(self, points, R, t, average=(), class_average=(), color=(), n_layers=None, alpha=0.5,
                        class_bayesian=(), bayesian_inference=()):
        """average / class_average / color / class_bayesian / bayesian_inference: lists of (cloud column, layer index).
        Uses the accepted-point counts of the frame just fused (self.last["cnt"] == new_elmap plane 2)."""
        pts, C = _pts(points), self.C
        R = np.ascontiguousarray(R, np.float32).ravel(); t = np.ascontiguousarray(t, np.float32)
        need = 1 + max([l for _, l in list(average) + list(class_average) + list(color) + list(class_bayesian) + list(bayesian_inference)] + [-1])
        n_layers = max(n_layers or 0, need)
        if not hasattr(self, "semantic_map") or self.semantic_map.shape[0] < n_layers:
            old = getattr(self, "semantic_map", np.zeros((0, C, C), np.float32))
            self.semantic_map = np.concatenate([old, np.zeros((n_layers - old.shape[0], C, C), np.float32)], axis=0)
        sm, cnt = self.semantic_map, self.last["cnt"]
        n, st = ct.c_long(pts.shape[0]), ct.c_long(pts.shape[1])
        for group, kind in ((average, 0), (class_average, 1)):
            if not group:
                continue
            ch = np.array([c for c, _ in group], np.int32); ly = np.array([l for _, l in group], np.int32)
            sums = np.zeros((sm.shape[0], C, C))
            lib().eo_sem_sum(ct.byref(self.P), _p(pts), n, st, _p(R), _p(t), ct.c_int(len(group)), _p(ch), _p(ly), _p(sums))
            if kind == 0:
                lib().eo_sem_average(ct.byref(self.P), _p(sums), _p(cnt), ct.c_int(len(group)), _p(ly), _p(sm))
            else:
                lib().eo_sem_class_average(ct.byref(self.P), _p(sums), _p(cnt), ct.c_int(len(group)), _p(ly), ct.c_double(alpha), _p(sm))
        if class_bayesian:
            if not hasattr(self, "semantic_alpha") or self.semantic_alpha.shape[0] < sm.shape[0]:
                old = getattr(self, "semantic_alpha", np.zeros((0, C, C), np.float32))
                self.semantic_alpha = np.concatenate([old, np.zeros((sm.shape[0] - old.shape[0], C, C), np.float32)], axis=0)
            ch = np.array([c for c, _ in class_bayesian], np.int32); ly = np.array([l for _, l in class_bayesian], np.int32)
            lib().eo_sem_class_bayesian(ct.byref(self.P), _p(pts), n, st, _p(R), _p(t), ct.c_int(len(ch)), _p(ch), _p(ly),
                                        _p(self.semantic_alpha), _p(sm))
        if bayesian_inference:
            ch = np.array([c for c, _ in bayesian_inference], np.int32); ly = np.array([l for _, l in bayesian_inference], np.int32)
            lib().eo_sem_bayesian_inference(ct.byref(self.P), _p(pts), n, st, _p(R), _p(t), _p(cnt), ct.c_int(len(ch)), _p(ch), _p(ly), _p(sm))
        assert len(color) <= 1, "oracle restates the single-colour-channel case (K>1 is a reference launch-size quirk)"
        for c_, l_ in color:
            lib().eo_sem_color(ct.byref(self.P), _p(pts), n, st, _p(R), _p(t), ct.c_int(c_), ct.c_int(l_), _p(sm))

    # ---- map shift (reference elevation_mapping.py:139-226): host array code, restated in NumPy ----------------------------
    def move_to(self, position):
        """:154-170 (the base rotation is not map state)"""
        if not hasattr(self, "center"):
            self.center = np.zeros(3, np.float32)
        position = np.asarray(position, np.float64)
        delta = position - self.center
        delta_pixel = np.around(delta[:2] / self.P.resolution)
        self.center[:2] += delta_pixel * self.P.resolution
        self.center[2] += delta[2]
        self.shift_map_xy(-delta_pixel)
        self.shift_map_z(-delta[2])

    def move(self, delta_position):
        """:139-152 (shifts by +delta_pixel: the sign differs from move_to, as in the reference)"""
        if not hasattr(self, "center"):
            self.center = np.zeros(3, np.float32)
        d = np.asarray(delta_position, np.float64)
        delta_pixel = np.round(d[:2] / self.P.resolution)
        self.center[:2] += delta_pixel * self.P.resolution
        self.center[2] += d[2]
        self.shift_map_xy(delta_pixel)
        self.shift_map_z(-d[2])

    def shift_map_xy(self, delta_pixel):
        """:200-214: roll by the integer shift, entering band = 0 on every plane, initial_variance on plane 1.  The normal map
        and traversability_input are NOT shifted (Appendix C of SURVEY.md)."""
        sr, sc = (int(v) for v in np.asarray(delta_pixel).astype(np.int32))
        if abs(sr) + abs(sc) == 0:
            return
        m = np.roll(self.elevation_map, (sr, sc), axis=(1, 2))
        for plane, value in ((slice(None), 0.0), (1, np.float32(self.P.initial_variance))):
            if sr > 0:
                m[plane, :sr, :] = value
            elif sr < 0:
                m[plane, sr:, :] = value
            if sc > 0:
                m[plane, :, :sc] = value
            elif sc < 0:
                m[plane, :, sc:] = value
        self.elevation_map = np.ascontiguousarray(m)
        if hasattr(self, "semantic_map"):
            sm = np.roll(self.semantic_map, (sr, sc), axis=(1, 2))
            if sr > 0:
                sm[:, :sr, :] = 0
            elif sr < 0:
                sm[:, sr:, :] = 0
            if sc > 0:
                sm[:, :, :sc] = 0
            elif sc < 0:
                sm[:, :, sc:] = 0
            self.semantic_map = np.ascontiguousarray(sm)

    def shift_map_z(self, delta_z):
        """:216-226; delta_z is a float64 scalar array in the reference: the sum is rounded once to float32"""
        self.elevation_map[0] += np.float64(delta_z)
        self.elevation_map[5] += np.float64(delta_z)

    def update_variance(self):
        lib().eo_update_variance(ct.byref(self.P), _p(self.elevation_map))

    def update_time(self):
        lib().eo_update_time(ct.byref(self.P), _p(self.elevation_map))

    # ---- whole frame: same sequence as reference update_map_with_kernel (elevation_mapping.py:316-391)
    def update_map_with_kernel(self, points, R, t, position_noise=0.0, orientation_noise=0.0):
        """``t`` must already be map-centre relative (the reference does ``t -= center`` first)."""
        self.count(points, R, t)
        self.gate(position_noise, orientation_noise)
        self.fuse(points, R, t)
        self.commit()
        if self.P.enable_visibility_cleanup:
            self.rays(points, R, t)
        self.average()
        if self.P.enable_overlap_clearance:
            self.overlap_clear(float(np.float32(np.asarray(t, np.float32)[2])))
        self.dilate()
        self.traversability()
        self.normals()

    def frame_c(self, points, R, t, position_noise=0.0, orientation_noise=0.0):
        """Whole frame inside one C call (used for cpu_baseline timing). Returns EoStats."""
        pts = _pts(points)
        R = np.ascontiguousarray(R, np.float32).ravel(); t = np.ascontiguousarray(t, np.float32)
        st = EoStats()
        lib().eo_frame(ct.byref(self.P), _p(self.elevation_map), _p(self.normal_map), _p(self.traversability_input),
                       _p(pts), ct.c_long(pts.shape[0]), ct.c_long(pts.shape[1]), _p(R), _p(t),
                       ct.c_double(position_noise), ct.c_double(orientation_noise), ct.byref(st))
        if st.gate_fired:
            self.mean_error = np.float32(st.mean_error)
            self.additive_mean_error = np.float32(self.additive_mean_error + np.float32(st.mean_error))
        return st


def image_correspondence(P, emap, x1, y1, z1, Pm, K, D, image_height, image_width, center):
    """reference image_to_map_correspondence_kernel; returns (uv (2,C,C) float32, valid (C,C) uint8)"""
    C = P.cell_n
    uv, valid = np.zeros((2, C, C), np.float32), np.zeros((C, C), np.uint8)
    f = ct.c_float
    a = lambda x: _p(np.ascontiguousarray(x, np.float32))  # noqa: E731
    lib().eo_image_correspondence(ct.byref(P), a(emap), f(x1), f(y1), f(z1), a(Pm), a(K), a(D), f(image_height), f(image_width),
                                  a(center), _p(uv), _p(valid))
    return uv, valid


def image_fuse(P, kind, sem_plane, image, uv, valid, image_height, image_width, alpha=0.7):
    """in-place update of one semantic plane; kind 'exponential' / 'average' (image (H,W)) or 'color' (image (3,H,W))"""
    img = np.ascontiguousarray(image, np.float32)
    lib().eo_image_fuse(ct.byref(P), ct.c_int({"exponential": 0, "color": 1, "average": 2}[kind]), _p(sem_plane), _p(img), _p(uv), _p(valid),
                        ct.c_float(image_height), ct.c_float(image_width), ct.c_double(alpha))


def polygon_mask(P, polygon, center_x, center_y):
    """0/1 mask of polygon_mask_kernel; polygon (M, 2) float32 world coordinates (already clipped to the map)"""
    poly = np.ascontiguousarray(polygon, np.float32)
    bbox = np.concatenate([poly.min(axis=0), poly.max(axis=0)]).astype(np.float32)
    mask = np.zeros((P.cell_n, P.cell_n), np.float32)
    lib().eo_polygon_mask(ct.byref(P), _p(poly), ct.c_int(poly.shape[0]), ct.c_float(center_x), ct.c_float(center_y), _p(bbox), _p(mask))
    return mask


def set_threads(n):
    """OpenMP threads of the C oracle (1 = sequential, bit-reproducible: the setting every parity test uses)."""
    lib().eo_set_threads(ct.c_int(int(n)))


def dilate_plane(C, d, plane, mask):
    out = np.zeros((C, C), np.float32); om = np.zeros((C, C), np.float32)
    lib().eo_dilate(ct.c_int(C), ct.c_int(d), _p(np.ascontiguousarray(plane, np.float32)),
                    _p(np.ascontiguousarray(mask, np.float32)), _p(out), _p(om))
    return out, om


def max_filter(C, d, iteration_n, elevation, valid):
    out = np.zeros((C, C), np.float32)
    n = lib().eo_max_filter(ct.c_int(C), ct.c_int(d), ct.c_int(iteration_n), _p(np.ascontiguousarray(elevation, np.float32)),
                            _p(np.ascontiguousarray(valid, np.float32)), _p(out))
    return out, int(n)


def smooth_filter(plane, passes=2):
    """SmoothFilter plugin (reference plugins/smooth_filter.py:56-58): the reference calls cupyx.scipy.ndimage.uniform_filter, whose
    published algorithm is scipy.ndimage.uniform_filter (third-party, version unpinned): scipy here is the oracle."""
    from scipy import ndimage
    h = np.ascontiguousarray(plane, np.float32)
    for _ in range(passes):
        h = ndimage.uniform_filter(h, size=3)
    return h


def min_filter(C, d, iteration_n, elevation, valid):
    out = np.zeros((C, C), np.float32)
    n = lib().eo_min_filter(ct.c_int(C), ct.c_int(d), ct.c_int(iteration_n), _p(np.ascontiguousarray(elevation, np.float32)),
                            _p(np.ascontiguousarray(valid, np.float32)), _p(out))
    return out, n
