"""``ElevationMap`` -- host-side mirror of the reference's map object for the point-cloud hot path.

Same method names / argument meaning as the reference class (reference EM/elevation_mapping.py:49-922) so the
ROS wrapper (src/elevation_mapping_wrapper.cpp:173-252) and the reference's tests drive it unchanged; the map
itself lives in HBM inside ``libemap_hip.so`` (C ABI ``include/emap_hip.h``).  NumPy is used for host arrays only;
there is no CuPy, no PyTorch and no CPU compute fallback here.
"""
from __future__ import annotations

import ctypes as ct
import os
import threading
from typing import List

import numpy as np

from . import _lib
from ._lib import EmapError, EmapParams, EmapStats, EmapStrip, PLANES, f32p
from .parameter import Parameter


def _shoelace_area(xy):
    """area of a simple polygon (n, 2)"""
    x, y = np.asarray(xy, np.float64).T
    return 0.5 * abs(float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))))


def _hull_ring(cells):
    """closed counter-clockwise hull ring (k + 1, 2) around integer cells (n, 2), or None when they do not span an area.  The
    reference asks shapely for ``MultiPoint(...).convex_hull`` (absent here); scipy's Qhull yields the same vertex set."""
    if cells.shape[0] < 3:
        return None
    from scipy.spatial import ConvexHull, QhullError
    pts = cells.astype(np.float64)
    try:
        v = ConvexHull(pts).vertices
    except QhullError:              # collinear cells
        return None
    return np.vstack([pts[v], pts[v[:1]]])


class LazyPlanes:
    """What a plugin receives as ``elevation_map`` / ``semantic_map``: behaves like the reference's ``(L, rows, C)`` array -- ``shape``,
    ``dtype``, ``ndim``, indexing, iteration, arithmetic, ``copy()`` and every other ndarray attribute work -- but a plane only crosses
    PCIe when the plugin actually touches it (the built-in plugins run on the device and never do).  Integer indexing fetches one
    plane; anything else materialises the stack once and delegates to NumPy."""

    def __init__(self, n, fetch, device_map=None, rows=None, cols=None):
        self._n, self._fetch, self._cache, self._full = int(n), fetch, {}, None
        if rows is None and device_map is not None:
            rows, cols = device_map.rows, device_map.cell_n
        self.shape = (self._n, int(rows or 0), int(cols or 0))
        self.dtype = np.dtype(np.float32)
        self.ndim = 3
        self.device_map = device_map          # the ElevationMap whose live core planes these are (built-in plugins read them on the device)

    def __len__(self):
        return self._n

    def _plane(self, k):
        k = int(k)
        if not -self._n <= k < self._n:
            raise IndexError("layer index %d out of range for %d layers" % (k, self._n))
        k %= self._n
        if k not in self._cache:
            self._cache[k] = self._fetch(k)
        return self._cache[k]

    def __getitem__(self, key):
        if isinstance(key, (int, np.integer)):
            return self._plane(key)
        return np.asarray(self)[key]

    def __iter__(self):
        return (self._plane(k) for k in range(self._n))

    def __array__(self, dtype=None, copy=None):
        if self._full is None:
            self._full = (np.stack([self._plane(k) for k in range(self._n)], axis=0) if self._n
                          else np.zeros(self.shape, np.float32))
        return self._full.astype(dtype) if dtype is not None else self._full

    def __getattr__(self, name):              # copy, sum, mean, astype, T, ...: whatever an ndarray offers
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(np.asarray(self), name)

    __hash__ = None


def _forward_operators():
    """arithmetic and comparisons of a LazyPlanes act on the materialised stack, like the reference's cupy array would"""
    def binary(name):
        return lambda self, other: getattr(np.asarray(self), name)(np.asarray(other) if isinstance(other, LazyPlanes) else other)

    def unary(name):
        return lambda self: getattr(np.asarray(self), name)()
    for o in ("add", "radd", "sub", "rsub", "mul", "rmul", "truediv", "rtruediv", "pow", "lt", "le", "gt", "ge", "eq", "ne"):
        setattr(LazyPlanes, "__%s__" % o, binary("__%s__" % o))
    for o in ("neg", "abs"):
        setattr(LazyPlanes, "__%s__" % o, unary("__%s__" % o))


_forward_operators()


class ElevationMap:
    """Core elevation mapping class (MI355X backend)."""

    layer_names_core = ["elevation", "variance", "is_valid", "traversability", "time", "upper_bound", "is_upper_bound"]

    def __init__(self, param: Parameter, strip=None, stream=None):
        """
        Args:
            param: ``Parameter`` (``update()`` is called if ``cell_n`` is unset).
            strip: optional ``(row_begin, row_count, halo_rows)`` for row-strip multi-GPU contexts.
            stream: optional raw ``hipStream_t`` (int) to enqueue on, e.g. ``torch.cuda.current_stream().cuda_stream``.
        """
        if param.cell_n is None:
            param.update()
        self.param = param
        self.data_type = np.float32
        self.resolution = param.resolution
        self.center = np.zeros(3, dtype=np.float32)
        self.base_rotation = np.eye(3, dtype=np.float32)
        self.map_length = param.map_length
        self.cell_n = int(param.cell_n)
        self.map_lock = threading.Lock()
        self.layer_names = list(self.layer_names_core)
        self.initial_variance = param.initial_variance
        self.mean_error = 0.0
        self.additive_mean_error = 0.0
        mode = param.index_mode
        if mode == "auto":
            mode = "reference_fp16" if self.cell_n <= 2049 else "fp32"
        self.index_mode = mode

        # traversability weights: same file format as the reference (elevation_mapping.py:103-104)
        wf = os.path.expandvars(os.path.expanduser(param.weight_file or ""))
        if wf and os.path.isfile(wf):
            param.load_weights(wf)

        # overlap clearance window (reference :88-91) -- recomputed inside the library, kept for API parity
        cell_range = int(np.clip(int(param.overlap_clear_range_xy / self.resolution), 0, self.cell_n))
        self.cell_min = self.cell_n // 2 - cell_range // 2
        self.cell_max = self.cell_n // 2 + cell_range // 2

        self._lib = _lib.load()
        self._strip = None
        if strip is not None:
            self._strip = EmapStrip(int(strip[0]), int(strip[1]), int(strip[2]), 0)
        self.rows = self.cell_n if strip is None else int(strip[1])
        self.row_begin = 0 if strip is None else int(strip[0])
        self._ctx = ct.c_void_p()
        P = _lib.fill_params(param, self.cell_n, mode)
        rc = self._lib.emap_create(ct.byref(P), ct.byref(self._strip) if self._strip is not None else None,
                                   int(param.device), ct.c_void_p(stream or 0), ct.byref(self._ctx))
        if rc != 0:
            self._ctx = ct.c_void_p()
            raise EmapError("emap_create failed with status %d (no usable HIP device / invalid parameters); "
                            "the MI355X backend has no CPU fallback" % rc)
        self._params_struct = P
        self.traversability_buffer = np.full((self.rows, self.cell_n), np.nan, np.float32)

        # managers are optional layers on top of the hot path (built lazily; see semantic_map / plugins)
        self.semantic_map = None
        self.plugin_manager = None
        from .semantic_map import SemanticMap
        self.semantic_map = SemanticMap(self.param, self)
        # plugins (reference :110-113): same YAML format; a missing file just means "no plugins"
        from .plugins.plugin_manager import PluginManager
        self.plugin_manager = PluginManager(cell_n=self.cell_n, emap=self)
        pf = os.path.expandvars(os.path.expanduser(param.plugin_config_file or ""))
        if pf and os.path.isfile(pf):
            self.plugin_manager.load_plugin_settings(pf)

    # ------------------------------------------------------------------------------------------------
    def _chk(self, rc):
        if rc != 0:
            raise EmapError("libemap_hip call failed (%d): %s" % (rc, self._lib.emap_last_error(self._ctx).decode()))

    def __del__(self):
        try:
            if getattr(self, "_ctx", None) and self._ctx.value:
                self._lib.emap_destroy(self._ctx)
                self._ctx = ct.c_void_p()
        except Exception:
            pass

    def close(self):
        self.__del__()

    def set_scatter_mode(self, mode, bin_stack=0):
        """"auto" | "atomic" | "binned": how count/fuse scatter into the map (bit-identical results).
        ``bin_stack`` (test hook) forces bins of that many stacked 16x64 tiles, as maps beyond 16384 tiles use."""
        self._chk(self._lib.emap_set_scatter_mode(self._ctx, {"auto": 0, "atomic": 1, "binned": 2}[mode] | (int(bin_stack) << 8)))
        self._scatter_mode = mode

    def last_update_path(self):
        """which kernels the last whole frame ran: "atomic" (chain of launches), "binned" (tile kernels) or "small_frame" (one launch,
        robot scale) -- include/emap_hip.h: emap_last_update_path; the results are bit-identical"""
        v = ct.c_int32(-1)
        self._chk(self._lib.emap_last_update_path(self._ctx, ct.byref(v)))
        return {0: "atomic", 1: "binned", 2: "small_frame"}[v.value & 3]

    def small_frame_aborts(self):
        """how many robot-scale frames were re-run on the chain of launches because their single launch aborted a grid barrier (foreign
        work held the GPU) -- include/emap_hip.h: emap_small_frame_aborts; normally 0, the results are the same either way"""
        v = ct.c_uint32(0)
        self._chk(self._lib.emap_small_frame_aborts(self._ctx, ct.byref(v)))
        return int(v.value)

    def last_frame_semantics(self):
        """how the last whole frame fused the channels declared for it (emap_frame_semantics): "in_tile_pass" (32-byte records, fused by
        the tile kernel that fused the heights), "carried" (32-byte records, the stand-alone semantic kernel read the channels from
        them: a launch with heavy-tile parts) or "separate" (the stand-alone kernels on 16-byte records / the atomic path)"""
        v = ct.c_int32(-1)
        self._chk(self._lib.emap_last_update_path(self._ctx, ct.byref(v)))
        return "in_tile_pass" if v.value & 4 else ("carried" if v.value & 8 else "separate")

    def set_ray_mode(self, mode):
        """"auto" | "by_row" | "by_ray": how a SHARDED frame (emap_update_sharded) runs the visibility pass (include/emap_hip.h:
        emap_set_ray_mode; bit-identical results, every rank the same setting)"""
        self._chk(self._lib.emap_set_ray_mode(self._ctx, {"auto": 0, "by_row": 1, "by_ray": 2}[mode]))
        self._ray_mode = {"auto": 0, "by_row": 1, "by_ray": 2}[mode]

    def reload_params(self):
        """Push changed ``self.param`` scalars to the device (kernargs, no recompilation)."""
        P = _lib.fill_params(self.param, self.cell_n, self.index_mode)
        self._chk(self._lib.emap_set_params(self._ctx, ct.byref(P)))
        self._params_struct = P

    # ---- state access ------------------------------------------------------------------------------
    def get_layer_raw(self, name_or_id):
        """Raw device plane as a host ``(rows, cell_n)`` float32 array."""
        pid = PLANES[name_or_id] if isinstance(name_or_id, str) else int(name_or_id)
        out = np.empty((self.rows, self.cell_n), np.float32)
        self._chk(self._lib.emap_get_layer(self._ctx, pid, f32p(out)))
        return out

    @property
    def logical_row_begin(self):
        """first LOGICAL map row of this context's (rows, cell_n) views: 0 for a full map; a strip holds the logical rows
        begin, begin + 1, ... modulo cell_n, and they change with every row shift (the strip keeps its physical rows)"""
        r = ct.c_int32(0)
        self._chk(self._lib.emap_strip_logical_begin(self._ctx, ct.byref(r)))
        return int(r.value)

    def set_layer_raw(self, name_or_id, array):
        pid = PLANES[name_or_id] if isinstance(name_or_id, str) else int(name_or_id)
        a = np.ascontiguousarray(array, np.float32)
        assert a.shape == (self.rows, self.cell_n), a.shape
        self._chk(self._lib.emap_set_layer(self._ctx, pid, f32p(a)))

    @property
    def elevation_map(self):
        """Host copy of the planar ``(7, cell_n, cell_n)`` map (reference attribute, elevation_mapping.py:67)."""
        return np.stack([self.get_layer_raw(k) for k in range(7)], axis=0)

    @elevation_map.setter
    def elevation_map(self, value):
        value = np.asarray(value, np.float32)
        for k in range(7):
            self.set_layer_raw(k, value[k])

    @property
    def normal_map(self):
        return np.stack([self.get_layer_raw(k) for k in (7, 8, 9)], axis=0)

    @normal_map.setter
    def normal_map(self, value):
        value = np.asarray(value, np.float32)
        for j, k in enumerate((7, 8, 9)):
            self.set_layer_raw(k, value[j])

    @property
    def traversability_input(self):
        return self.get_layer_raw(10)

    # ---- reference API -----------------------------------------------------------------------------
    def clear(self):
        """Reset all layers (reference :119-128)."""
        with self.map_lock:
            self._chk(self._lib.emap_clear(self._ctx))
            if self.semantic_map is not None:
                self.semantic_map.clear()
        self.mean_error = 0.0
        self.additive_mean_error = 0.0

    def get_position(self, position):
        position[0][:] = self.center

    def move(self, delta_position):
        """Relative shift (reference :139-152 -- note the sign convention differs from move_to).  The caller's float64 position
        meets the float32 centre in float64, exactly like the reference's array arithmetic (pinned by tests/golden/host_steps.npz)."""
        delta_position = np.asarray(delta_position, np.float64)
        delta_pixel = np.round(delta_position[:2] / self.resolution)
        self.center[:2] += delta_pixel * self.resolution
        self.center[2] += delta_position[2]
        self._shift(delta_pixel, -float(delta_position[2]))

    def move_to(self, position, R):
        """Absolute shift + base rotation update (reference :154-170)."""
        self.base_rotation = np.asarray(R, dtype=np.float32)
        position = np.asarray(position, np.float64)
        delta = position - self.center
        delta_pixel = np.around(delta[:2] / self.resolution)
        self.center[:2] += delta_pixel * self.resolution
        self.center[2] += delta[2]
        self._shift(-delta_pixel, -float(delta[2]))

    def _shift(self, delta_pixel, dz):
        sv = np.asarray(delta_pixel).astype(np.int32)
        with self.map_lock:
            self._chk(self._lib.emap_shift(self._ctx, int(sv[0]), int(sv[1]), ct.c_float(dz)))
            if self.semantic_map is not None and np.abs(sv).sum() != 0:
                self.semantic_map.shift_map_xy(sv)

    def shift_map_xy(self, delta_pixel):
        self._shift(delta_pixel, 0.0)

    def shift_map_z(self, delta_z):
        self._shift(np.zeros(2), float(delta_z))

    def shift_translation_to_map_center(self, t):
        t -= self.center

    def bind_points(self, points_all, strip_pose=None):
        """Upload a host cloud ``(N, 3+K)`` (float32 or float64) and bind it for the next stage calls.  ``strip_pose = (R, t)`` (t
        map-centre relative) on a row-strip context: only the points that can land in this strip's rows under that pose are converted
        and uploaded (include/emap_hip.h: emap_upload_points_strip -- the frame must then use the same pose and must not march its
        rays by row); returns the number of points bound."""
        pts = np.asarray(points_all)
        if pts.dtype == np.float64:
            pts = np.ascontiguousarray(pts)
            dtype = 1
        else:
            pts = np.ascontiguousarray(pts, np.float32)
            dtype = 0
        assert pts.ndim == 2 and pts.shape[1] >= 3
        if strip_pose is not None:
            R, t = self._rt(*strip_pose)
            kept = ct.c_int64(0)
            self._chk(self._lib.emap_upload_points_strip(self._ctx, ct.c_void_p(pts.ctypes.data), ct.c_int64(pts.shape[0]),
                                                         ct.c_int64(pts.shape[1]), dtype, f32p(R), f32p(t), ct.byref(kept)))
            self._n_bound = int(kept.value)
            self._bound_host = None                 # (the bound cloud is a subset: nothing indexes the host copy by point)
            return self._n_bound
        self._chk(self._lib.emap_upload_points(self._ctx, ct.c_void_p(pts.ctypes.data), ct.c_int64(pts.shape[0]),
                                               ct.c_int64(pts.shape[1]), dtype))
        self._n_bound = pts.shape[0]
        self._bound_host = pts
        return self._n_bound

    def strip_point_mask(self, points_all, R, t):
        """which points ``bind_points(points_all, strip_pose=(R, t))`` would upload (bool array; all True on a whole-map context)"""
        pts = np.asarray(points_all)
        dtype = 1 if pts.dtype == np.float64 else 0
        pts = np.ascontiguousarray(pts) if dtype else np.ascontiguousarray(pts, np.float32)
        R, t = self._rt(R, t)
        keep = np.zeros(pts.shape[0], np.uint8)
        self._chk(self._lib.emap_strip_point_mask(self._ctx, ct.c_void_p(pts.ctypes.data), ct.c_int64(pts.shape[0]), ct.c_int64(pts.shape[1]), dtype,
                                                  f32p(R), f32p(t), keep.ctypes.data_as(ct.POINTER(ct.c_uint8))))
        return keep.astype(bool)

    def bind_points_device(self, dev_ptr, n, stride):
        """Bind a device-resident float32 cloud (raw pointer) without copying."""
        self._chk(self._lib.emap_set_points_device(self._ctx, ct.c_void_p(dev_ptr), ct.c_int64(n), ct.c_int64(stride)))
        self._n_bound = n
        self._bound_host = None

    def bind_points_device_split(self, xyz_ptr, chan_ptr, n, n_chan):
        """Bind a device-resident cloud that is already de-interleaved: xyz ``(n, 3)`` and the extra channels ``(n, n_chan)``, both
        row-major float32 (the layout ``bind_points`` / ``input_pointcloud`` give an uploaded cloud)."""
        self._chk(self._lib.emap_set_points_device_split(self._ctx, ct.c_void_p(xyz_ptr), ct.c_void_p(chan_ptr), ct.c_int64(n), ct.c_int64(n_chan)))
        self._n_bound = n
        self._bound_host = None

    @staticmethod
    def _rt(R, t):
        R = np.ascontiguousarray(np.asarray(R, np.float32).reshape(9))
        t = np.ascontiguousarray(np.asarray(t, np.float32).reshape(3))
        return R, t

    def update_map_with_kernel(self, points_all, channels, R, t, position_noise, orientation_noise, want_stats=True):
        """One frame (reference :316-391).  ``points_all``: host ``(N, 3+K)`` array or ``None`` to reuse the bound
        cloud; ``t`` is shifted to the map centre in place like the reference does (:333)."""
        if points_all is not None:
            self.bind_points(points_all)
        R, t32 = self._rt(R, t)
        with self.map_lock:
            jobs = []
            if self.semantic_map is not None and channels:
                # layers + count plane must exist before the average pass; the sum / colour fusions ride inside the frame
                jobs = self.semantic_map.declare_frame(self, list(channels))
            t32 = t32 - self.center
            try:
                t -= self.center  # reference mutates the caller's array
            except TypeError:
                pass
            st = EmapStats()
            self._chk(self._lib.emap_update(self._ctx, f32p(R), f32p(t32), ct.c_double(position_noise),
                                            ct.c_double(orientation_noise), ct.byref(st) if want_stats else None))
            if want_stats:
                self._take_stats(st)
            if jobs:
                self.semantic_map.finish_frame(self, jobs, R, t32)
        return st if want_stats else None

    def _take_stats(self, st):
        if st.gate_fired:
            self.mean_error = float(st.mean_error)
        self.additive_mean_error = float(st.additive_mean_error)
        self.last_stats = st

    def clear_overlap_map(self, t):
        self._chk(self._lib.emap_overlap_clear(self._ctx, ct.c_float(float(np.asarray(t, np.float32).reshape(3)[2]))))

    def get_additive_mean_error(self):
        st = EmapStats()
        self._chk(self._lib.emap_get_stats(self._ctx, ct.byref(st)))
        self.additive_mean_error = float(st.additive_mean_error)
        return self.additive_mean_error

    def update_variance(self):
        self._chk(self._lib.emap_update_variance(self._ctx))

    def update_time(self):
        self._chk(self._lib.emap_update_time(self._ctx))

    def update_normal(self, dilated_map):
        """normal map from a given (dilated) height plane and the current ``is_valid`` (reference :564-577: ``normal_map *= 0`` + the
        normal filter kernel).  Inside a frame the stencil launch does this on the device planes; this public form takes a host plane:
        it becomes the stencil stage's input, the stage runs, and the two planes the reference's method does not touch (the
        traversability it would also rewrite, the stage's input plane) are put back."""
        with self.map_lock:
            keep_trav, keep_in = self.get_layer_raw("traversability"), self.get_layer_raw("traversability_input")
            try:
                self.set_layer_raw("traversability_input", np.asarray(dilated_map, np.float32))
                self.stage("traversability_normals")
            finally:                                  # whatever the stage did: the two planes the reference's method leaves alone come back
                self.set_layer_raw("traversability", keep_trav)
                self.set_layer_raw("traversability_input", keep_in)

    def update_upper_bound_with_valid_elevation(self):
        m = self.elevation_map
        mask = m[2] > 0.5
        self.set_layer_raw(5, np.where(mask, m[0], m[5]))
        self.set_layer_raw(6, np.where(mask, 0.0, m[6]))

    def input_pointcloud(self, raw_points, channels: List[str], R, t, position_noise: float, orientation_noise: float):
        """Entry point used by the ROS wrapper (reference :434-466).  NaN rows are skipped inside the kernels
        instead of being compacted on the host (:458)."""
        additional_channels = list(channels[3:])
        # asynchronous: nothing is read back per frame (get_additive_mean_error / get_map_with_name_ref synchronise)
        self.update_map_with_kernel(raw_points, additional_channels, np.asarray(R, np.float32),
                                    np.asarray(t, np.float32).copy(), position_noise, orientation_noise, want_stats=False)

    def input_image(self, image, channels: List[str], R, t, K, D, distortion_model: str, image_height: int, image_width: int):
        """Fuse a (multi-channel) camera image into the semantic layers (reference :468-562): every known cell is
        projected into the image, occluded cells are rejected by a Bresenham walk towards the camera, the image is
        sampled per cell."""
        image = np.stack([np.asarray(c) for c in image], axis=0)
        if image.ndim == 2:
            image = image[None]
        image = image.astype(np.float32)
        K = np.asarray(K, np.float32); R = np.asarray(R, np.float32); t = np.asarray(t, np.float32)
        D = np.asarray(D, np.float32).reshape(-1)
        if len(D) < 4:
            D = np.zeros(5, np.float32)
        elif len(D) == 4:
            D = np.concatenate([D, np.zeros(1, np.float32)])
        else:
            D = D[:5].copy()
        if distortion_model != "radtan":
            D = D * 0            # equidistant / plumb_bob: "not implemented yet" in the reference -> no distortion
        P = (K @ np.concatenate([R, t[:, None]], 1)).astype(np.float32)
        t_cam_map = -R.T @ t - self.center
        x1 = np.float32(np.uint32((self.cell_n / 2) + (t_cam_map[0] / self.resolution)))
        y1 = np.float32(np.uint32((self.cell_n / 2) + (t_cam_map[1] / self.resolution)))
        z1 = np.float32(t_cam_map[2])
        with self.map_lock:
            self._chk(self._lib.emap_image_correspondence(
                self._ctx, ct.c_float(x1), ct.c_float(y1), ct.c_float(z1), f32p(np.ascontiguousarray(P.reshape(-1))),
                f32p(np.ascontiguousarray(K.reshape(-1))), f32p(np.ascontiguousarray(D)), ct.c_float(float(image_height)),
                ct.c_float(float(image_width)), f32p(np.ascontiguousarray(self.center, np.float32))))
            self.semantic_map.update_layers_image(self, image, list(channels), image_height, image_width)

    def get_image_correspondence(self):
        """(uv (2, C, C) float32, valid (C, C) bool) of the last input_image call"""
        uv = np.empty((2, self.cell_n, self.cell_n), np.float32); valid = np.empty((self.cell_n, self.cell_n), np.uint8)
        self._chk(self._lib.emap_image_get_correspondence(self._ctx, f32p(uv), valid.ctypes.data_as(ct.c_void_p)))
        return uv, valid.astype(bool)

    input = input_pointcloud  # name used by the C++ wrapper (src/elevation_mapping_wrapper.cpp:173-178)

    # ---- stage-level API (parity tests; same order as update_map_with_kernel) ----------------------
    def stage(self, name, R=None, t=None, **kw):
        L = self._lib
        if name in ("count", "fuse", "rays", "fuse_average"):
            R, t = self._rt(R, t)
            self._chk(getattr(L, "emap_" + name)(self._ctx, f32p(R), f32p(t)))
        elif name == "gate":
            s, c = kw.get("err_sum"), kw.get("err_cnt")
            self._chk(L.emap_set_drift_inputs(self._ctx, ct.c_double(kw.get("position_noise", 0.0)),
                                              ct.c_double(kw.get("orientation_noise", 0.0)),
                                              ct.byref(ct.c_double(s)) if s is not None else None,
                                              ct.byref(ct.c_uint32(c)) if c is not None else None))
        elif name == "overlap":
            self._chk(L.emap_overlap_clear(self._ctx, ct.c_float(float(t))))
        elif name in ("commit", "average", "dilate", "traversability_normals", "post", "update_variance", "update_time"):
            self._chk(getattr(L, "emap_" + name)(self._ctx))
        else:
            raise ValueError(name)

    def stats(self):
        st = EmapStats()
        self._chk(self._lib.emap_get_stats(self._ctx, ct.byref(st)))
        return st

    def point_index(self, R, t):
        """(idx, valid, inside) per bound point -- tail of add_points_kernel (custom_kernels.py:260-262)."""
        R, t = self._rt(R, t)
        n = self._n_bound
        idx, valid, inside = np.empty(n, np.int32), np.empty(n, np.uint8), np.empty(n, np.uint8)
        self._chk(self._lib.emap_point_index(self._ctx, f32p(R), f32p(t), idx.ctypes.data_as(ct.c_void_p),
                                             valid.ctypes.data_as(ct.c_void_p), inside.ctypes.data_as(ct.c_void_p)))
        return idx, valid, inside

    def sync(self):
        self._chk(self._lib.emap_sync(self._ctx))

    # ---- read-back (reference :579-775) ----------------------------------------------------------
    def exists_layer(self, name):
        if name in self.layer_names:
            return True
        if self.semantic_map is not None and name in self.semantic_map.layer_names:
            return True
        if self.plugin_manager is not None and name in self.plugin_manager.layer_names:
            return True
        return False

    def _publish(self, m, fill_nan=False, add_z=False, valid=None):
        m = m.copy()
        if fill_nan:
            m = np.where(valid > 0.5, m, np.nan)
        if add_z:
            m = m + self.center[2]
        return m[1:-1, 1:-1]

    def get_map_with_name_ref(self, name, data):
        """Fill ``data`` (float32, ``(cell_n-2, cell_n-2)``) in place: border stripped, both axes flipped
        (reference :720-775)."""
        builtin = {"elevation": 0, "variance": 1, "traversability": 2, "time": 3, "upper_bound": 4, "is_upper_bound": 5,
                   "normal_x": 6, "normal_y": 7, "normal_z": 8}
        if name in builtin and self._strip is None and isinstance(data, np.ndarray) and data.dtype == np.float32 \
                and data.flags["C_CONTIGUOUS"] and data.shape == (self.cell_n - 2, self.cell_n - 2):
            with self.map_lock:   # strip / NaN fill / +center_z / double flip happen in one device kernel
                self._chk(self._lib.emap_publish_layer(self._ctx, builtin[name], ct.c_float(float(self.center[2])),
                                                       int(bool(self.param.use_only_above_for_upper_bound)), f32p(data)))
            return
        with self.map_lock:
            m = self._stripped_layer(name)
        if m is None:
            print("Layer {} is not in the map".format(name))
            return
        m = np.flip(np.flip(m, 0), 1)
        data[...] = m.astype(np.float32)

    def _require_full_map(self, what):
        if self._strip is not None:
            raise EmapError("%s works on a full map; this context holds the row strip [%d, %d) -- gather the strips first "
                            "(ShardedElevationMap.gather(name) assembles the full plane on every rank)" % (what, self.row_begin, self.row_begin + self.rows))

    def _stripped_layer(self, name):
        """border-stripped, unflipped layer as the reference's get_* accessors return it (:598-680, :740-765)"""
        self._require_full_map("layer read-back (%s)" % name)
        if name == "elevation":
            return self._publish(self.get_layer_raw(0), True, True, self.get_layer_raw(2))
        if name == "variance":
            return self._publish(self.get_layer_raw(1))
        if name == "traversability":
            trav = np.where((self.get_layer_raw(2) + self.get_layer_raw(6)) > 0.5, self.get_layer_raw(3), np.nan)
            self.traversability_buffer[3:-3, 3:-3] = trav[3:-3, 3:-3]
            return self.traversability_buffer[1:-1, 1:-1]
        if name == "time":
            return self._publish(self.get_layer_raw(4))
        if name in ("upper_bound", "is_upper_bound"):
            e = self.elevation_map
            if self.param.use_only_above_for_upper_bound:
                valid = np.logical_or(np.logical_and(e[5] > 0.0, e[6] > 0.5), e[2] > 0.5)
            else:
                valid = np.logical_or(e[2] > 0.5, e[6] > 0.5)
            if name == "upper_bound":
                return np.where(valid, e[5], np.nan)[1:-1, 1:-1] + self.center[2]
            return np.where(valid, e[6], np.nan)[1:-1, 1:-1]
        if name in ("normal_x", "normal_y", "normal_z"):
            return self.get_layer_raw(name)[1:-1, 1:-1]
        if self.semantic_map is not None and name in self.semantic_map.layer_names:
            return self.semantic_map.get_map_with_name(name)
        if self.plugin_manager is not None and name in self.plugin_manager.layer_names:
            sem = self.semantic_map
            self.plugin_manager.update_with_name(
                name, LazyPlanes(7, self.get_layer_raw, device_map=self), self.layer_names,
                LazyPlanes(len(sem.layer_names), sem._layer, rows=self.rows, cols=self.cell_n) if sem is not None else None,
                self.semantic_map.layer_names if self.semantic_map is not None else [],
                self.base_rotation, self.semantic_map.elements_to_shift if self.semantic_map is not None else {})
            m = self.plugin_manager.get_map_with_name(name)
            p = self.plugin_manager.get_param_with_name(name)
            return self._publish(np.asarray(m), p.fill_nan, p.is_height_layer, self.get_layer_raw(2))
        return None

    # the reference's per-layer accessors (:598-680): host arrays, border stripped, NOT flipped
    def process_map_for_publish(self, input_map, fill_nan=False, add_z=False, xp=np):
        return self._publish(np.asarray(input_map), fill_nan, add_z, self.get_layer_raw(2) if fill_nan else None)

    def get_elevation(self):
        return self._stripped_layer("elevation")

    def get_variance(self):
        return self._stripped_layer("variance")

    def get_traversability(self):
        return self._stripped_layer("traversability")

    def get_time(self):
        return self._stripped_layer("time")

    def get_upper_bound(self):
        return self._stripped_layer("upper_bound")

    def get_is_upper_bound(self):
        return self._stripped_layer("is_upper_bound")

    def get_normal_maps(self):
        """(3, C-2, C-2), flipped on both axes like the reference (:776-790)"""
        n = self.normal_map[:, 1:-1, 1:-1]
        return np.ascontiguousarray(np.flip(np.flip(n, 1), 2))

    def xp_of_array(self, array):
        return np if isinstance(array, np.ndarray) else None

    def copy_to_cpu(self, array, data, stream=None):
        data[...] = np.asarray(array).astype(np.float32)

    # ---- safety-polygon service (reference :837-897) ---------------------------------------------------------
    def polygon_mask(self, polygon):
        """(cell_n, cell_n) 0/1 mask of the cells inside ``polygon`` ((M, 2) world coordinates, already clipped): the device
        kernel behind get_polygon_traversability (polygon_mask_kernel, custom_kernels.py:509-651)."""
        poly = np.ascontiguousarray(polygon, np.float32)
        mask = np.empty((self.cell_n, self.cell_n), np.float32)
        self._chk(self._lib.emap_polygon_mask(self._ctx, f32p(poly), int(poly.shape[0]), ct.c_float(float(self.center[0])),
                                              ct.c_float(float(self.center[1])), f32p(mask)))
        return mask

    def get_polygon_traversability(self, polygon, result):
        """Safety check of a footprint polygon (node service; reference :837-889).  The cell mask comes from the device
        (``emap_polygon_mask`` = the reference's polygon_mask_kernel, bit-exact); the statistics are a few reductions over the
        masked interior: ``result`` = (is_safe, mean untraversability of the known cells inside, polygon area); the return value is
        the vertex count of the hull around the cells that are too untraversable (``get_untraversable_polygon`` hands it out)."""
        self._require_full_map("get_polygon_traversability")
        poly = np.asarray(polygon, np.float64)
        area = _shoelace_area(poly)
        reach = self.map_length / 2 - self.resolution                  # the polygon is clipped to the map interior (:851-855)
        clipped = np.clip(poly.astype(np.float32), self.center[:2] - reach, self.center[:2] + reach).astype(np.float32)
        with self.map_lock:
            self.mask = self.polygon_mask(clipped)
            layer = np.asarray(self.get_layer(self.param.checker_layer), np.float32)
            valid = self.get_layer_raw("is_valid")
        inner = (slice(1, -1), slice(1, -1))
        footprint, known = self.mask[inner], valid[inner]
        risk = np.where(known > 0.5, 1.0 - layer[inner], 0.0) * footprint           # unknown cells count as traversable
        n_known = float((known * footprint).sum())
        mean_risk = float(risk.sum()) / n_known if n_known > 0 else 0.0
        too_risky = risk > 1.0 - self.param.safe_thresh
        safe = int(too_risky.sum()) <= self.param.max_unsafe_n and float(risk.max()) <= 1.0 - self.param.safe_min_thresh
        if _shoelace_area(clipped) < 0.001:
            print("requested polygon is outside of the map")
            safe = False
        ring = _hull_ring(np.argwhere(too_risky))
        if ring is not None:                                            # cell indices of the border-stripped view -> map frame
            ring = self.center[:2].reshape(1, 2) + (ring - self.cell_n / 2.0) * self.resolution
        self.untraversable_polygon = ring
        result[...] = np.array([safe, mean_risk, area])
        return 0 if ring is None else int(ring.shape[0])

    def get_untraversable_polygon(self, untraversable_polygon):
        untraversable_polygon[...] = np.asarray(self.untraversable_polygon)

    # ---- map initialisation from a few known points (reference :899-923, map_initializer.py:25-62) -----------------
    def initialize_map(self, points, method="cubic"):
        """Interpolate the elevation between ``points`` ((M, 3) world x, y, z; e.g. the feet positions) with
        ``scipy.interpolate.griddata`` -- host code in the reference too -- then two dilation passes of radius
        ``dilation_size_initialize`` on the device and upper bound := elevation on valid cells."""
        from scipy.interpolate import griddata
        self._require_full_map("initialize_map")
        self.clear()
        with self.map_lock:
            points = np.array(points, dtype=np.float32)
            # world x, y -> (fractional, truncated) cell indices around the map centre
            indices = ((points[:, :2] - self.center[:2].astype(np.float32).reshape(1, 2)) / self.resolution + self.cell_n / 2).astype(np.int32)
            points[:, :2] = indices.astype(points.dtype)
            points[:, 2] -= self.center[2]
            m = self.elevation_map
            known = np.where(m[2] > 0.5)
            pts_idx = np.vstack([np.stack(known).T, points[:, :2]])
            values = np.hstack([m[0][known], points[:, 2]])
            assert pts_idx.shape[0] > 3, "Initialization points must be more than 3."
            gx, gy = np.mgrid[0:self.cell_n, 0:self.cell_n]
            interpolated = griddata(pts_idx, values, (gx, gy), method=method)
            ok = ~np.isnan(interpolated)
            m[0] = np.nan_to_num(interpolated)
            m[1] = np.where(ok, self.param.initialized_variance, self.initial_variance)
            m[2] = np.where(ok, 1.0, 0.0)
            if self.param.dilation_size_initialize > 0:
                e = np.ascontiguousarray(m[0], np.float32); v = np.ascontiguousarray(m[2], np.float32)
                oe, ov = np.empty_like(e), np.empty_like(v)
                self._chk(self._lib.emap_dilate_planes(self._ctx, f32p(e), f32p(v), int(self.param.dilation_size_initialize), 2, f32p(oe), f32p(ov)))
                m[0], m[2] = oe, ov
            mask = m[2] > 0.5                       # update_upper_bound_with_valid_elevation (:428-432)
            m[5] = np.where(mask, m[0], m[5])
            m[6] = np.where(mask, 0.0, m[6])
            for k in (0, 1, 2, 5, 6):
                self.set_layer_raw(k, m[k])

    def compile_kernels(self):
        """nothing to compile: parameters are kernel arguments of the prebuilt gfx950 library (reference :228-282)"""

    compile_image_kernels = compile_kernels

    def pad_value(self, x, shift_value, idx=None, value=0.0):
        """host helper of the reference's shift (:172-198): fill the band a shift vacated in a (planes, rows, columns) stack.  The map
        itself is shifted on the device (``emap_shift``: circular origin, no copy); this is for callers that shift their own arrays."""
        from .semantic_map import SemanticMap
        SemanticMap.pad_value(None, x, shift_value, idx=idx, value=value)

    def get_normal_ref(self, normal_x_data, normal_y_data, normal_z_data):
        n = self.get_normal_maps()
        normal_x_data[...] = n[0]
        normal_y_data[...] = n[1]
        normal_z_data[...] = n[2]

    def get_layer(self, name):
        """Raw (unflipped, unstripped) layer by name (reference :777-791)."""
        if name in self.layer_names:
            return self.get_layer_raw(self.layer_names.index(name))
        if self.semantic_map is not None and name in self.semantic_map.layer_names:
            return self.semantic_map._layer(self.semantic_map.layer_names.index(name))
        if self.plugin_manager is not None and name in self.plugin_manager.layer_names:
            sem = self.semantic_map
            self.plugin_manager.update_with_name(
                name, LazyPlanes(7, self.get_layer_raw, device_map=self), self.layer_names,
                LazyPlanes(len(sem.layer_names), sem._layer, rows=self.rows, cols=self.cell_n) if sem is not None else None,
                self.semantic_map.layer_names if self.semantic_map is not None else [],
                self.base_rotation, self.semantic_map.elements_to_shift if self.semantic_map is not None else {})
            return self.plugin_manager.get_map_with_name(name)
        print("Layer {} is not in the map, returning traversabiltiy!".format(name))
        return None
