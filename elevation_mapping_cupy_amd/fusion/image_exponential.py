"""Descriptor of the reference image fusion EM/fusion/image_exponential.py:40-77 (exponential_correspondences_to_map_kernel, alpha 0.7); arithmetic: csrc/emap_semantic.hip (k_image_fuse)."""
from .fusion_manager import FusionBase


class ImageExponential(FusionBase):
    def __init__(self, params, *args, **kwargs):
        self.name = "image_exponential"
        self.kind = "exponential"
        self.alpha = 0.7
        self.cell_n = params.cell_n
        self.resolution = params.resolution
