"""``Parameter`` -- plain-dataclass counterpart of the reference's parameter object
(reference EM/parameter.py:13-289; the C++ node fills it by reflection through get_names/get_types/set_value,
src/elevation_mapping_wrapper.cpp:45-77).  Same field names, defaults and helper methods; no simple_parsing.
Additional field ``index_mode`` selects the rounding mode of the HIP kernels.
"""
from __future__ import annotations

import pickle
from dataclasses import dataclass, field, fields

import numpy as np


def _zeros(*shape):
    return field(default_factory=lambda: np.zeros(shape, np.float32))


class _WeightsUnpickler(pickle.Unpickler):
    _ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"), ("numpy", "dtype"),
                ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"), ("collections", "OrderedDict"),
                ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
                ("_codecs", "encode")}          # protocol-2 pickles rebuild an array's bytes with _codecs.encode(str, "latin1")

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError("weight file refers to %s.%s: only numpy arrays in a dict are accepted" % (module, name))


@dataclass
class Parameter:
    resolution: float = 0.04
    subscriber_cfg: dict = field(default_factory=lambda: {
        "front_cam": {"channels": ["rgb", "person"], "topic_name": "/elevation_mapping/pointcloud_semantic",
                      "data_type": "pointcloud"}})
    additional_layers: list = field(default_factory=lambda: ["color"])
    fusion_algorithms: list = field(default_factory=lambda: [
        "image_color", "image_exponential", "pointcloud_average", "pointcloud_bayesian_inference",
        "pointcloud_class_average", "pointcloud_class_bayesian", "pointcloud_class_max", "pointcloud_color"])
    pointcloud_channel_fusions: dict = field(default_factory=lambda: {"rgb": "color", "default": "class_average"})
    image_channel_fusions: dict = field(default_factory=lambda: {"rgb": "color", "default": "exponential"})
    data_type: str = np.float32
    average_weight: float = 0.5

    map_length: float = 8.0
    sensor_noise_factor: float = 0.05
    mahalanobis_thresh: float = 2.0
    outlier_variance: float = 0.01
    drift_compensation_variance_inlier: float = 0.1
    time_variance: float = 0.01
    time_interval: float = 0.1

    max_variance: float = 1.0
    dilation_size: float = 2
    dilation_size_initialize: float = 10
    drift_compensation_alpha: float = 1.0

    traversability_inlier: float = 0.1
    wall_num_thresh: float = 100
    min_height_drift_cnt: float = 100

    max_ray_length: float = 2.0
    cleanup_step: float = 0.01
    cleanup_cos_thresh: float = 0.5
    min_valid_distance: float = 0.3
    max_height_range: float = 1.0
    ramped_height_range_a: float = 0.3
    ramped_height_range_b: float = 1.0
    ramped_height_range_c: float = 0.2

    safe_thresh: float = 0.5
    safe_min_thresh: float = 0.5
    max_unsafe_n: int = 20
    checker_layer: str = "traversability"

    min_filter_size: int = 5
    min_filter_iteration: int = 3

    max_drift: float = 0.10

    overlap_clear_range_xy: float = 4.0
    overlap_clear_range_z: float = 2.0

    enable_edge_sharpen: bool = True
    enable_drift_compensation: bool = True
    enable_visibility_cleanup: bool = True
    enable_overlap_clearance: bool = True
    use_only_above_for_upper_bound: bool = True
    use_chainer: bool = True  # kept for ROS-parameter compatibility; ignored (the filter is a HIP kernel)
    position_noise_thresh: float = 0.1
    orientation_noise_thresh: float = 0.1

    plugin_config_file: str = "config/plugin_config.yaml"
    weight_file: str = "config/weights.dat"

    initial_variance: float = 10.0
    initialized_variance: float = 10.0
    w1: np.ndarray = _zeros(4, 1, 3, 3)
    w2: np.ndarray = _zeros(4, 1, 3, 3)
    w3: np.ndarray = _zeros(4, 1, 3, 3)
    w_out: np.ndarray = _zeros(1, 12, 1, 1)

    true_map_length: float = None
    cell_n: int = None
    true_cell_n: int = None

    # MI355X backend knobs (not in the reference)
    index_mode: str = "auto"   # "reference_fp16" | "fp32" | "auto" (fp16 quirk when cell_n <= 2049)
    device: int = 0

    def load_weights(self, filename):
        """Same file format as the reference (pickled dict of numpy arrays, parameter.py:227-238).  A pickle can run arbitrary
        code when it is opened: only the classes such a file needs (numpy array reconstruction, dict / OrderedDict) are resolved."""
        with open(filename, "rb") as f:
            weights = _WeightsUnpickler(f).load()
        self.w1 = np.asarray(weights["conv1.weight"], np.float32)
        self.w2 = np.asarray(weights["conv2.weight"], np.float32)
        self.w3 = np.asarray(weights["conv3.weight"], np.float32)
        self.w_out = np.asarray(weights["conv_final.weight"], np.float32)

    def get_names(self):
        return [f.name for f in fields(self)]

    def get_types(self):
        return [getattr(f.type, "__name__", str(f.type)) if not isinstance(f.type, str) else f.type.split(".")[-1]
                for f in fields(self)]

    def set_value(self, name, value):
        setattr(self, name, value)

    def get_value(self, name):
        return getattr(self, name)

    def update(self):
        # +2: one border cell on each side (reference parameter.py:282-289)
        self.cell_n = int(round(self.map_length / self.resolution)) + 2
        self.true_cell_n = round(self.map_length / self.resolution)
        self.true_map_length = self.true_cell_n * self.resolution
