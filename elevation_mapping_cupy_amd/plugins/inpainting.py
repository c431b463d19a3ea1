"""``Inpainting`` plugin (reference EM/plugins/inpainting.py:14-63).

Same semantics around the fill as the reference: mask = ``is_valid < 0.5``, the elevation is quantised to 8 bits over
[h_min, h_max] of the valid cells (``astype("uint8")`` truncation), inpainted with radius 1, de-quantised
(``* (h_max - h_min) / 255 + h_min``), returned as float64.  The fill itself:

* ``method="telea"`` (the reference's default, ``cv2.INPAINT_TELEA``): Telea's fast-marching method, HOST code of ``libemap_hip.so``
  (``emap_inpaint_telea_u8``) -- the reference runs this step on the CPU as well (``cp.asnumpy`` + ``cv2.inpaint``): the algorithm is
  a serial priority-queue march.  OpenCV is an unpinned third-party dependency that is absent here, so this is a restatement of the
  published algorithm and **parity with OpenCV's values is unpinned** (tests/test_inpaint_telea.py pins it against a second,
  line-by-line restatement and checks the properties every implementation must have).
* ``method="front"``: the round-1/2 device-side substitute (``emap_inpaint_u8``: front-by-front distance-weighted mean of the known
  8-neighbours on the MI355X) for maps where a host pass per publish is too slow; needs the owning ElevationMap.
* ``method="ns"`` (``cv2.INPAINT_NS``, reference plugins/inpainting.py:33-38): the Navier-Stokes based method in its fast-marching form,
  HOST code as well (``emap_inpaint_ns_u8``): the same march as Telea's, a pixel = the mean of the known pixels within the radius
  weighted along the isophote direction.  Parity with OpenCV's values unpinned like "telea" (tests/test_inpaint_ns.py)."""
from __future__ import annotations

import ctypes as ct
from typing import List

import numpy as np

from .. import _lib
from .._lib import f32p
from .plugin_manager import PluginBase


class Inpainting(PluginBase):
    def __init__(self, cell_n: int = 100, method: str = "telea", emap=None, **kwargs):
        super().__init__()
        self.method = method if method in ("telea", "ns", "front") else "telea"      # (the reference falls back to telea for unknown names, :37-38)
        self.cell_n = cell_n
        self.emap = emap
        self.sweeps_run = 0

    def __call__(self, elevation_map: np.ndarray, layer_names: List[str], plugin_layers: np.ndarray,
                 plugin_layer_names: List[str], *args) -> np.ndarray:
        known = np.ascontiguousarray(np.asarray(elevation_map[2]) >= 0.5)
        if not known.any() or known.all():
            return elevation_map[0]
        h = np.asarray(elevation_map[0], np.float32)
        h_max, h_min = float(h[known].max()), float(h[known].min())
        span = (h_max - h_min) if h_max > h_min else 1.0
        q8 = np.clip((h - h_min) * 255 / span, 0, 255).astype(np.uint8)            # 8-bit image, truncation like astype("uint8")
        if self.method in ("telea", "ns"):
            lib = _lib.load()
            mask = np.ascontiguousarray(~known, np.uint8)
            out8 = np.empty_like(q8)
            p = lambda a: a.ctypes.data_as(ct.POINTER(ct.c_uint8))               # noqa: E731
            fill = lib.emap_inpaint_telea_u8 if self.method == "telea" else lib.emap_inpaint_ns_u8
            rc = fill(p(np.ascontiguousarray(q8)), p(mask), q8.shape[0], q8.shape[1], 1, p(out8))
            if rc != 0:
                raise _lib.EmapError("emap_inpaint_%s_u8 failed (%d)" % (self.method, rc))
            out = out8.astype(np.float32)
        else:
            if self.emap is None:
                raise RuntimeError("Inpainting(method='front') needs the owning ElevationMap (PluginManager(emap=...)): it runs on the device")
            q = q8.astype(np.float32)
            out = np.empty_like(q)
            n = ct.c_int32(0)
            e = self.emap
            e._chk(e._lib.emap_inpaint_u8(e._ctx, f32p(np.ascontiguousarray(q)), f32p(known.astype(np.float32)),
                                          int(2 * self.cell_n), f32p(out), ct.byref(n)))
            self.sweeps_run = n.value
        return (out * np.float32(span) / np.float32(255) + np.float32(h_min)).astype(np.float64)
