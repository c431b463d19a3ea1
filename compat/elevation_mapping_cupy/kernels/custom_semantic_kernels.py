"""Names of EM/kernels/custom_semantic_kernels.py.  The reference splits every semantic fusion into an accumulate kernel that sums
into float planes (``sum_kernel``, ``sum_compact_kernel``, ``sum_max_kernel``, ``add_color_kernel``, ``alpha_kernel``) and a finalise
kernel that reads them (``average_kernel``, ``class_average_kernel``, ``bayesian_inference_kernel``, ``color_average_kernel``).  The
MI355X library fuses each pair into one tile kernel whose accumulators live in LDS (``emap_semantic_update``; fp64 / uint32, never
written to HBM), so the intermediate planes these factories exchange do not exist here.  The fusions themselves are available --
bit-pinned against the reference kernels' outputs -- through the reference's own higher-level entry point
``SemanticMap.update_layers_pointcloud`` / ``FusionManager`` (``elevation_mapping_cupy.semantic_map``).  The factories below exist so
that ``from elevation_mapping_cupy.kernels import ...`` resolves; calling one explains the above."""


def _unavailable(name, fusion):
    def factory(*args, **kwargs):
        raise NotImplementedError(
            "%s: the accumulate / finalise split of the reference's semantic kernels has no counterpart in the MI355X library (one "
            "fused tile kernel per fusion, accumulators in LDS); use SemanticMap.update_layers_pointcloud with the '%s' fusion" % (name, fusion))
    factory.__name__ = name
    return factory


sum_kernel = _unavailable("sum_kernel", "average")
average_kernel = _unavailable("average_kernel", "average")
sum_compact_kernel = _unavailable("sum_compact_kernel", "class_average / class_bayesian")
class_average_kernel = _unavailable("class_average_kernel", "class_average")
alpha_kernel = _unavailable("alpha_kernel", "class_bayesian")
sum_max_kernel = _unavailable("sum_max_kernel", "class_max")
bayesian_inference_kernel = _unavailable("bayesian_inference_kernel", "bayesian_inference")
add_color_kernel = _unavailable("add_color_kernel", "color")
color_average_kernel = _unavailable("color_average_kernel", "color")
