"""Descriptor of the reference fusion EM/fusion/pointcloud_class_bayesian.py:56-75 (alpha_kernel: Dirichlet pseudo-counts summed per
cell into persistent layers, then theta = alpha / sum(alpha) over the fusion's layers); arithmetic: csrc/emap_semantic.hip (kind 2)."""
from .fusion_manager import FusionBase


class ClassBayesian(FusionBase):
    def __init__(self, params, *args, **kwargs):
        self.name = "pointcloud_class_bayesian"
        self.kind = "class_bayesian"
        self.cell_n = params.cell_n
        self.resolution = params.resolution
