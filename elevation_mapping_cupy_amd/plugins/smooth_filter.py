"""``SmoothFilter`` plugin: two 3x3 uniform-filter passes over a layer (reference EM/plugins/smooth_filter.py:12-59, which calls
``cupyx.scipy.ndimage.uniform_filter(h, size=3)`` twice).  Runs on the MI355X (``emap_smooth_filter``)."""
from __future__ import annotations

from typing import List

import numpy as np

from .._lib import f32p
from .plugin_manager import PluginBase


class SmoothFilter(PluginBase):
    def __init__(self, cell_n: int = 100, input_layer_name: str = "elevation", emap=None, **kwargs):
        super().__init__()
        self.input_layer_name = input_layer_name
        self.emap = emap

    def __call__(self, elevation_map: np.ndarray, layer_names: List[str], plugin_layers: np.ndarray,
                 plugin_layer_names: List[str], *args) -> np.ndarray:
        if self.emap is None:
            raise RuntimeError("SmoothFilter needs the owning ElevationMap (PluginManager(emap=...)): it runs on the device")
        if self.input_layer_name in layer_names:
            h = elevation_map[layer_names.index(self.input_layer_name)]
        elif self.input_layer_name in plugin_layer_names:
            h = plugin_layers[plugin_layer_names.index(self.input_layer_name)]
        else:
            print("layer name {} was not found. Using elevation layer.".format(self.input_layer_name))
            h = elevation_map[0]
        h = np.ascontiguousarray(h, np.float32)
        out = np.empty_like(h)
        e = self.emap
        e._chk(e._lib.emap_smooth_filter(e._ctx, f32p(h), 2, f32p(out)))
        return out
