"""Descriptor of the reference image fusion EM/fusion/image_color.py (color_correspondences_to_map_kernel); arithmetic: csrc/emap_semantic.hip (k_image_fuse)."""
from .fusion_manager import FusionBase


class ImageColor(FusionBase):
    def __init__(self, params, *args, **kwargs):
        self.name = "image_color"
        self.kind = "color"
        self.alpha = 0.7
        self.cell_n = params.cell_n
        self.resolution = params.resolution
