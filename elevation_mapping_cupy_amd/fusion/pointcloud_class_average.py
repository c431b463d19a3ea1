"""Descriptor of the reference fusion EM/fusion/pointcloud_class_average.py:106-126 (sum_kernel + class_average_kernel, alpha = average_weight); arithmetic: csrc/emap_semantic.hip."""
from .fusion_manager import FusionBase


class ClassAverage(FusionBase):
    def __init__(self, params, *args, **kwargs):
        self.name = "pointcloud_class_average"
        self.kind = "class_average"
        self.cell_n = params.cell_n
        self.resolution = params.resolution
