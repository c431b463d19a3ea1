"""``Inpainting`` plugin (reference EM/plugins/inpainting.py:14-63).

Same surrounding semantics as the reference: mask = ``is_valid < 0.5``, the elevation is quantised to 8 bits over
[h_min, h_max] of the valid cells, inpainted, de-quantised (``* (h_max - h_min) / 255 + h_min``).  The reference then
calls OpenCV's ``cv2.inpaint(h, mask, 1, INPAINT_TELEA)``; OpenCV is an unpinned third-party dependency that is absent
here, so the fill itself is a DOCUMENTED SUBSTITUTE that runs on the MI355X (``emap_inpaint_u8``: front-by-front
distance-weighted mean of the known 8-neighbours, 8-bit rounding).  Parity with OpenCV's values is therefore unpinned;
the tests check the properties any inpainting must have (valid cells reproduce the 8-bit round trip, filled values stay
inside the range of the known data, every cell is filled)."""
from __future__ import annotations

import ctypes as ct
from typing import List

import numpy as np

from .._lib import f32p
from .plugin_manager import PluginBase


class Inpainting(PluginBase):
    def __init__(self, cell_n: int = 100, method: str = "telea", emap=None, **kwargs):
        super().__init__()
        self.method = method          # kept for configuration compatibility; both names select the substitute
        self.cell_n = cell_n
        self.emap = emap
        self.sweeps_run = 0

    def __call__(self, elevation_map: np.ndarray, layer_names: List[str], plugin_layers: np.ndarray,
                 plugin_layer_names: List[str], *args) -> np.ndarray:
        if self.emap is None:
            raise RuntimeError("Inpainting needs the owning ElevationMap (PluginManager(emap=...)): it runs on the device")
        known = np.ascontiguousarray(elevation_map[2] >= 0.5)
        if not known.any():
            return elevation_map[0]
        h = np.asarray(elevation_map[0], np.float32)
        h_max, h_min = float(h[known].max()), float(h[known].min())
        span = (h_max - h_min) if h_max > h_min else 1.0
        q = np.clip((h - h_min) * 255 / span, 0, 255).astype(np.uint8).astype(np.float32)     # 8-bit image, truncation like astype("uint8")
        out = np.empty_like(q)
        n = ct.c_int32(0)
        e = self.emap
        e._chk(e._lib.emap_inpaint_u8(e._ctx, f32p(np.ascontiguousarray(q)), f32p(known.astype(np.float32)),
                                      int(2 * self.cell_n), f32p(out), ct.byref(n)))
        self.sweeps_run = n.value
        return (out * np.float32(span) / np.float32(255) + np.float32(h_min)).astype(np.float64)
