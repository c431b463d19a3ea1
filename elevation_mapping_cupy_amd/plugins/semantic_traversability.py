"""``SemanticTraversability`` plugin (reference EM/plugins/semantic_traversability.py:12-83): votes of named layers -- a layer of type
"traversability" votes where it is AT OR BELOW its threshold, any other type where it is at or above -- and the result is 1 where at
least one layer voted, 0.1 elsewhere.  A layer name that exists nowhere ends the plugin without a layer (the reference returns None)."""
from typing import List

import numpy as np

from .plugin_manager import PluginBase


class SemanticTraversability(PluginBase):
    def __init__(self, cell_n: int = 100, layers: list = ("traversability",), thresholds: list = (0.5,), type: list = ("traversability",), **kwargs):
        super().__init__()
        self.layers = list(layers)
        self.thresholds = np.asarray(thresholds, np.float64)
        self.type = list(type)

    def __call__(self, elevation_map, layer_names: List[str], plugin_layers, plugin_layer_names: List[str], semantic_map,
                 semantic_layer_names: List[str], *args):
        votes = np.zeros(np.asarray(elevation_map[2]).shape, np.float32)
        for it, name in enumerate(self.layers):
            if name in layer_names:
                layer = np.asarray(elevation_map[list(layer_names).index(name)])
            elif name in plugin_layer_names:
                layer = np.asarray(plugin_layers[list(plugin_layer_names).index(name)])
            else:
                print("Layer {} is not in the map, returning traversabiltiy!".format(name))
                return None
            hit = layer <= self.thresholds[it] if self.type[it] == "traversability" else layer >= self.thresholds[it]
            votes += hit
        return np.where(votes <= 0.9, 0.1, 1.0)
