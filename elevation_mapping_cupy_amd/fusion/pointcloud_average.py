"""Descriptor of the reference fusion EM/fusion/pointcloud_average.py:93-113 (sum_kernel + average_kernel); arithmetic: csrc/emap_semantic.hip."""
from .fusion_manager import FusionBase


class Average(FusionBase):
    def __init__(self, params, *args, **kwargs):
        self.name = "pointcloud_average"
        self.kind = "average"
        self.cell_n = params.cell_n
        self.resolution = params.resolution
