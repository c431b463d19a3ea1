"""Descriptor of the reference fusion EM/fusion/pointcloud_color.py:131-152 (add_color_kernel + color_average_kernel); arithmetic: csrc/emap_semantic.hip."""
from .fusion_manager import FusionBase


class Color(FusionBase):
    def __init__(self, params, *args, **kwargs):
        self.name = "pointcloud_color"
        self.kind = "color"
        self.cell_n = params.cell_n
        self.resolution = params.resolution
