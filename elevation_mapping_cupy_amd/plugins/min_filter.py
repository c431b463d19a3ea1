"""``MinFilter`` plugin: fill invalid cells with the minimum value around them (reference EM/plugins/min_filter.py:12-118).
The sweeps run on the MI355X (``emap_min_filter``: LDS-tiled, double buffered, no host sync between sweeps)."""
from __future__ import annotations

import ctypes as ct
from typing import List

import numpy as np

from .._lib import f32p
from .plugin_manager import PluginBase


class MinFilter(PluginBase):
    def __init__(self, cell_n: int = 100, dilation_size: int = 5, iteration_n: int = 5, emap=None, **kwargs):
        super().__init__()
        self.iteration_n = int(iteration_n)
        self.dilation_size = int(dilation_size)
        self.width = self.height = cell_n
        self.emap = emap
        self.sweeps_run = 0

    def __call__(self, elevation_map: np.ndarray, layer_names: List[str], plugin_layers: np.ndarray,
                 plugin_layer_names: List[str], *args) -> np.ndarray:
        if self.emap is None:
            raise RuntimeError("MinFilter needs the owning ElevationMap (PluginManager(emap=...)): it runs on the device")
        e = self.emap
        out = np.empty((e.cell_n, e.cell_n), np.float32)
        n = ct.c_int32(0)
        if getattr(elevation_map, "device_map", None) is e:
            # the map's own live planes (get_map_with_name_ref): NULL inputs = read elevation / is_valid on the device, only the
            # result crosses PCIe
            e._chk(e._lib.emap_min_filter(e._ctx, None, None, self.dilation_size, self.iteration_n, f32p(out), ct.byref(n)))
        else:
            h = np.ascontiguousarray(elevation_map[0], np.float32)
            v = np.ascontiguousarray(elevation_map[2], np.float32)
            e._chk(e._lib.emap_min_filter(e._ctx, f32p(h), f32p(v), self.dilation_size, self.iteration_n, f32p(out), ct.byref(n)))
        self.sweeps_run = n.value
        return out
