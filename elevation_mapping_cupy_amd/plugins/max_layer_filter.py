"""``MaxLayerFilter`` plugin (reference EM/plugins/max_layer_filter.py:12-108): per-cell maximum (or minimum) over a list of named
layers after an optional chain per layer -- zeros replaced by a default value or by another layer, ``1 - x``, a scale, a 0/1
threshold.  With none of the layers present: a constant plane of the default value, or the traversability layer."""
from typing import List

import numpy as np

from .plugin_manager import PluginBase


class MaxLayerFilter(PluginBase):
    def __init__(self, cell_n: int = 100, layers: list = ("traversability",), reverse: list = (True,), min_or_max: str = "max",
                 thresholds: list = (False,), scales: list = (1.0,), default_value: float = 0.0, **kwargs):
        super().__init__()
        self.layers, self.reverse, self.min_or_max = list(layers), list(reverse), min_or_max
        self.thresholds, self.scales, self.default_value = list(thresholds), list(scales), default_value

    def __call__(self, elevation_map, layer_names: List[str], plugin_layers, plugin_layer_names: List[str], semantic_map,
                 semantic_layer_names: List[str], *args) -> np.ndarray:
        look = (elevation_map, list(layer_names), plugin_layers, list(plugin_layer_names), semantic_map, list(semantic_layer_names))
        prepared = []
        for it, name in enumerate(self.layers):
            layer = self.get_layer_data(*look, name)
            if layer is None:
                continue
            layer = np.asarray(layer)
            if isinstance(self.default_value, float):
                layer = np.where(layer == 0.0, float(self.default_value), layer)
            elif isinstance(self.default_value, str):
                layer = np.where(layer == 0, self.get_layer_data(*look, self.default_value), layer)
            if self.reverse[it]:
                layer = 1.0 - layer
            if len(self.scales) > it and isinstance(self.scales[it], float):
                layer = layer * float(self.scales[it])
            if isinstance(self.thresholds[it], float):
                layer = np.where(layer > float(self.thresholds[it]), 1, 0)
            prepared.append(layer)
        if not prepared:
            print("No layers are found, returning traversability!")
            if isinstance(self.default_value, float):
                return np.full_like(np.asarray(elevation_map[0]), float(self.default_value))
            return elevation_map[list(layer_names).index("traversability")]
        stack = np.stack(prepared, axis=0)
        return stack.min(axis=0) if self.min_or_max == "min" else stack.max(axis=0)
