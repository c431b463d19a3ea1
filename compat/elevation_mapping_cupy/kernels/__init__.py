"""Kernel-factory surface of the reference (``elevation_mapping_cupy.kernels``, EM/kernels/__init__.py) on the MI355X library.

The reference builds CuPy ``ElementwiseKernel`` objects from strings; a factory call returns ``k(*arrays, size=n)``.  Here a factory
returns a callable with the same argument list that takes NumPy arrays (the stand-in for CuPy arrays), runs the matching staged
C-ABI calls of ``libemap_hip.so`` on a scratch context of the factory's map size and writes the outputs back IN PLACE, as the
reference kernels do.  See ``custom_kernels.py`` for what maps one-to-one and what cannot (the deterministic fixed-point
accumulators of this implementation replace the reference's float ``new_map``)."""
from .custom_kernels import (add_points_kernel, error_counting_kernel, average_map_kernel, dilation_filter_kernel,  # noqa: F401
                             normal_filter_kernel, polygon_mask_kernel)
from .custom_image_kernels import (image_to_map_correspondence_kernel, average_correspondences_to_map_kernel,  # noqa: F401
                                   exponential_correspondences_to_map_kernel, color_correspondences_to_map_kernel)
from .custom_semantic_kernels import (sum_kernel, sum_compact_kernel, sum_max_kernel, alpha_kernel, average_kernel,  # noqa: F401
                                      bayesian_inference_kernel, class_average_kernel, add_color_kernel, color_average_kernel)
