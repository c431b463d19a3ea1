"""Parameter sets of the reference as plain dicts + a helper that builds a ``Parameter`` for a given map size.

CORE_PARAM_YAML restates the values of the reference's shipped ``elevation_mapping_cupy/config/core/core_param.yaml``
(the BASELINE workloads use them); the dataclass defaults live in ``parameter.py``."""
from __future__ import annotations

from .parameter import Parameter

CORE_PARAM_YAML = dict(
    resolution=0.04, map_length=8.0, sensor_noise_factor=0.05, mahalanobis_thresh=2.0, outlier_variance=0.01,
    drift_compensation_alpha=0.1, max_drift=0.1, time_variance=0.0001, max_variance=100.0, initial_variance=1000.0,
    traversability_inlier=0.9, dilation_size=3, wall_num_thresh=20.0, min_height_drift_cnt=100.0,
    position_noise_thresh=0.01, orientation_noise_thresh=0.01, min_valid_distance=0.5, max_height_range=1.0,
    ramped_height_range_a=0.3, ramped_height_range_b=1.0, ramped_height_range_c=0.2, time_interval=0.1,
    max_ray_length=10.0, cleanup_step=0.1, cleanup_cos_thresh=0.1, overlap_clear_range_xy=4.0,
    overlap_clear_range_z=2.0, enable_edge_sharpen=True, enable_visibility_cleanup=True,
    enable_drift_compensation=True, enable_overlap_clearance=True,
)


def parameter_from(cfg=None, cell_n=None, index_mode="auto", weights=None, device=0):
    """``Parameter`` with ``cfg`` overrides; ``cell_n`` (incl. border) fixes map_length = (cell_n-2)*resolution."""
    p = Parameter()
    for k, v in (cfg or {}).items():
        if hasattr(p, k):
            setattr(p, k, v)
    if cell_n is not None:
        p.map_length = (cell_n - 2) * p.resolution
    p.update()
    if cell_n is not None and p.cell_n != cell_n:
        raise ValueError("cannot realise cell_n=%d at resolution %g" % (cell_n, p.resolution))
    p.index_mode = index_mode
    p.weight_file = ""
    p.device = device
    if weights is not None:
        p.w1, p.w2, p.w3, p.w_out = weights["w1"], weights["w2"], weights["w3"], weights["w_out"]
    return p
