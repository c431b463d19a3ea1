"""``Erosion`` plugin (reference EM/plugins/erosion.py:12-113): quantise a layer to 8 bits over its value range, erode it with a
``kernel_size`` x ``kernel_size`` rectangle ``iterations`` times, de-quantise; ``reverse`` erodes ``1 - layer``.  The reference
calls ``cv2.erode`` (OpenCV: third-party, unpinned, absent here); ``emap_erode`` computes what that call is defined to compute
(window minimum, pixels outside the image ignored) on the MI355X."""
from __future__ import annotations

from typing import List

import numpy as np

from .._lib import f32p
from .plugin_manager import PluginBase


class Erosion(PluginBase):
    def __init__(self, input_layer_name="traversability", kernel_size: int = 3, iterations: int = 1, reverse: bool = False,
                 default_layer_name: str = "traversability", emap=None, **kwargs):
        super().__init__()
        self.input_layer_name = input_layer_name
        self.kernel_size = int(kernel_size)
        self.iterations = int(iterations)
        self.reverse = bool(reverse)
        self.default_layer_name = default_layer_name
        self.emap = emap

    def __call__(self, elevation_map: np.ndarray, layer_names: List[str], plugin_layers: np.ndarray, plugin_layer_names: List[str],
                 semantic_map: np.ndarray, semantic_layer_names: List[str], *args) -> np.ndarray:
        if self.emap is None:
            raise RuntimeError("Erosion needs the owning ElevationMap (PluginManager(emap=...)): it runs on the device")
        layer = None
        for name in (self.input_layer_name, self.default_layer_name, "traversability"):
            layer = self.get_layer_data(elevation_map, layer_names, plugin_layers, plugin_layer_names, semantic_map,
                                        semantic_layer_names, name)
            if layer is not None:
                break
            print(f"No layers are found, using {self.default_layer_name}!")
        layer = np.asarray(layer, np.float32)
        if self.reverse:
            layer = 1 - layer
        lo, hi = float(layer.min()), float(layer.max())
        if not hi > lo:                         # flat layer: nothing to erode (the reference divides by zero here)
            return (1 - layer) if self.reverse else layer
        q = ((layer - lo) * 255 / (hi - lo)).astype("uint8").astype(np.float32)
        out = np.empty_like(q)
        e = self.emap
        e._chk(e._lib.emap_erode(e._ctx, f32p(np.ascontiguousarray(q)), self.kernel_size, self.iterations, f32p(out)))
        out = out.astype(np.float32) * (hi - lo) / 255 + lo
        if self.reverse:
            out = 1 - out
        return out.astype(np.float32)
