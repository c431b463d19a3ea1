"""``SemanticFilter`` plugin (reference EM/plugins/semantic_filter.py:13-133): among the layers whose names match one of the
``classes`` patterns (elevation, plugin and semantic layers, in that order) take the per-cell argmax and return its colour from the
PASCAL-VOC style colour table, packed 0x00RRGGBB and bit-cast to float32 -- the class map the node publishes as an RGB layer."""
import re
from typing import List

import numpy as np

from .plugin_manager import PluginBase


def _voc_colors(n: int = 255) -> np.ndarray:
    """(n, 3) uint8: colour i (1-based) = bits of i dealt round-robin to r, g, b from the top bit down; entries 1..3 overridden like the
    reference's table (:60-62)"""
    idx = np.arange(n + 1, dtype=np.uint32)
    rgb = np.zeros((n + 1, 3), np.uint32)
    for j in range(8):
        for ch in range(3):
            rgb[:, ch] |= ((idx >> np.uint32(3 * j + ch)) & np.uint32(1)) << np.uint32(7 - j)
    rgb[1] = rgb[2] = (81, 113, 162)
    rgb[3] = (188, 63, 59)
    return rgb[1:].astype(np.uint8)


class SemanticFilter(PluginBase):
    def __init__(self, cell_n: int = 100, classes: list = ("person", "grass"), **kwargs):
        super().__init__()
        self.classes = list(classes)
        c = _voc_colors(255).astype(np.uint32)
        self.color_encoding = ((c[:, 0] << np.uint32(16)) | (c[:, 1] << np.uint32(8)) | c[:, 2]).view(np.float32)

    def get_layer_indices(self, layer_names: List[str]) -> List[int]:
        return [i for i, name in enumerate(layer_names) if any(re.match(pattern, name) for pattern in self.classes)]

    def __call__(self, elevation_map, layer_names: List[str], plugin_layers, plugin_layer_names: List[str], semantic_map,
                 semantic_layer_names: List[str], rotation=None, elements_to_shift=None, *args) -> np.ndarray:
        picked = []
        for stack, names in ((elevation_map, layer_names), (plugin_layers, plugin_layer_names), (semantic_map, semantic_layer_names)):
            idx = self.get_layer_indices(list(names))
            if idx:
                picked.append(np.asarray(stack)[idx])
        if picked:
            class_id = np.argmax(np.concatenate(picked, axis=0), axis=0)
        else:
            class_id = np.zeros(np.asarray(elevation_map[0]).shape, np.int64)
        return self.color_encoding[class_id]
