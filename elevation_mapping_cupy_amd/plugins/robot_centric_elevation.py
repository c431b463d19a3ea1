"""``RobotCentricElevation`` plugin (reference EM/plugins/robot_centric_elevation.py:12-121): the height of every valid cell seen from
the robot's base frame -- the cell's (row * resolution, column * resolution, elevation) rotated by the base rotation, z component --
or, with ``use_threshold``, 1 / 0 for at / below ``threshold``; invalid cells keep their elevation value.  A publish-time layer:
one vectorised NumPy expression on the host copies the plugin manager hands out (float32 arithmetic in the reference's order)."""
from typing import List

import numpy as np

from .plugin_manager import PluginBase


class RobotCentricElevation(PluginBase):
    def __init__(self, cell_n: int = 100, resolution: float = 0.05, threshold: float = 0.4, use_threshold: bool = False, **kwargs):
        super().__init__()
        self.cell_n = int(cell_n)
        self.threshold = np.float32(threshold)
        self.use_threshold = bool(use_threshold)
        # get_map_x / get_map_y (:46-53): integer cell index times the resolution, evaluated in float
        cells = np.arange(self.cell_n, dtype=np.float32) * np.float32(resolution)
        self._rx = cells[:, None]
        self._ry = cells[None, :]
        self.min_filtered = np.zeros((self.cell_n, self.cell_n), np.float32)

    def __call__(self, elevation_map, layer_names: List[str], plugin_layers, plugin_layer_names: List[str], semantic_map,
                 semantic_layer_names: List[str], rotation, *args) -> np.ndarray:
        h = np.asarray(elevation_map[0], np.float32)
        valid = np.asarray(elevation_map[2]) > 0.5
        R = np.asarray(rotation, np.float32).reshape(-1)
        z_b = (R[6] * self._rx + R[7] * self._ry) + R[8] * h                      # transform_p (:54-57), third row
        out = h.copy()
        if self.use_threshold:
            out[valid] = (z_b >= self.threshold).astype(np.float32)[valid]
        else:
            out[valid] = z_b[valid]
        self.min_filtered = out
        return out
