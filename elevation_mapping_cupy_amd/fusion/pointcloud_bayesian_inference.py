"""Descriptor of the reference fusion EM/fusion/pointcloud_bayesian_inference.py:86-122 (sum_compact_kernel + bayesian_inference_kernel).
The reference keeps the prior variance in ``new_map`` layers that semantic_map.py:243 zeroes before every fusion, so its posterior
mean equals the previous layer value; the device code restates the formula literally (csrc/emap_semantic.hip, kind 3)."""
from .fusion_manager import FusionBase


class BayesianInference(FusionBase):
    def __init__(self, params, *args, **kwargs):
        self.name = "pointcloud_bayesian_inference"
        self.kind = "bayesian_inference"
        self.cell_n = params.cell_n
        self.resolution = params.resolution
