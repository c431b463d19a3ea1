"""``FeaturesPca`` plugin (reference EM/plugins/features_pca.py:14-96): the layers whose names match ``process_layer_names`` (clipped
to [-1, 1]) are the feature vector of a cell; its first three principal components, each scaled to 0..255 over the map, make the
packed 0x00RRGGBB colour of the cell.  Host code in the reference too (``.get()`` + scikit-learn); the PCA here is the same
definition (centred data, SVD, components' signs fixed like scikit-learn's ``svd_flip``) without the dependency.

Which scikit-learn this matches: the sign convention is the v-based ``svd_flip`` of scikit-learn >= 1.5 with its exact
("full") solver; older releases flip by the LEFT singular vectors and pick the randomized solver above 500 samples, so a component --
and with it a colour channel -- can come out inverted relative to such an environment (the reference pins no version).  Fewer than
three matching layers: the missing components are zero columns here, scikit-learn raises."""
import re
from typing import List

import numpy as np

from .plugin_manager import PluginBase


def _pca3(data: np.ndarray) -> np.ndarray:
    """(n, f) -> (n, 3) scores on the first three principal axes (fewer features: padded with zero columns)"""
    x = data - data.mean(axis=0)
    u, s, vt = np.linalg.svd(x, full_matrices=False)
    # deterministic signs: the entry of largest magnitude in each right-singular vector is positive (sklearn.utils.extmath.svd_flip, v-based)
    flip = np.sign(vt[np.arange(vt.shape[0]), np.argmax(np.abs(vt), axis=1)])
    flip[flip == 0] = 1.0
    scores = (u * s) * flip
    out = np.zeros((data.shape[0], 3), scores.dtype)
    k = min(3, scores.shape[1])
    out[:, :k] = scores[:, :k]
    return out


class FeaturesPca(PluginBase):
    def __init__(self, cell_n: int = 100, process_layer_names: List[str] = (), **kwargs):
        super().__init__()
        self.process_layer_names = list(process_layer_names)

    def get_layer_indices(self, layer_names: List[str]) -> List[int]:
        return [i for i, name in enumerate(layer_names) if any(re.match(pattern, name) for pattern in self.process_layer_names)]

    def __call__(self, elevation_map, layer_names: List[str], plugin_layers, plugin_layer_names: List[str], semantic_map,
                 semantic_layer_names: List[str], *args) -> np.ndarray:
        shape = np.asarray(elevation_map[0]).shape
        cols = []
        for stack, names in ((elevation_map, layer_names), (plugin_layers, plugin_layer_names), (semantic_map, semantic_layer_names)):
            idx = self.get_layer_indices(list(names))
            if idx:
                cols.append(np.clip(np.asarray(stack)[idx].reshape(len(idx), -1).T, -1, 1))
        if not cols:
            return np.zeros(shape, np.float32)
        comp = _pca3(np.concatenate(cols, axis=1).astype(np.float64)).reshape(shape[0], shape[1], 3)
        lo, hi = comp.min(axis=(0, 1)), comp.max(axis=(0, 1))
        span = np.where(hi > lo, hi - lo, 1.0)
        img = ((comp - lo) / span * 255).astype(np.uint8).astype(np.uint32)
        return ((img[:, :, 0] << np.uint32(16)) | (img[:, :, 1] << np.uint32(8)) | img[:, :, 2]).view(np.float32)
