"""GPU: seeded random sweep over map sizes, parameter values, poses, cloud shapes and both scatter paths -- every
frame through emap_update, compared with the oracle's update_map_with_kernel plane by plane."""
import numpy as np
import pytest

import _fixtures as fx
from _util import assert_planes_close, make_pair
from oracle import emap_oracle as eo

pytestmark = pytest.mark.gpu


def _random_case(seed):
    rng = np.random.default_rng(seed)
    C = int(rng.choice([34, 66, 98, 130, 202, 258, 322]))
    cfg = dict(eo.DEFAULTS if rng.uniform() < 0.5 else eo.YAML)
    cfg.update(
        sensor_noise_factor=float(rng.choice([0.01, 0.05, 0.2])), mahalanobis_thresh=float(rng.choice([1.0, 2.0, 3.5])),
        outlier_variance=float(rng.choice([0.001, 0.01, 0.1])), wall_num_thresh=float(rng.choice([3, 20, 100])),
        max_ray_length=float(rng.choice([1.0, 2.0, 6.0])), cleanup_step=float(rng.choice([0.01, 0.1])),
        cleanup_cos_thresh=float(rng.choice([0.1, 0.5])), min_valid_distance=float(rng.choice([0.0, 0.3, 0.5])),
        max_height_range=float(rng.choice([0.5, 1.0])), ramped_height_range_a=float(rng.choice([0.1, 0.3])),
        max_variance=float(rng.choice([1.0, 100.0])), initial_variance=float(rng.choice([10.0, 1000.0])),
        dilation_size=int(rng.choice([1, 2, 3, 5])), enable_edge_sharpen=bool(rng.integers(2)),
        enable_visibility_cleanup=bool(rng.integers(2)), enable_overlap_clearance=bool(rng.integers(2)),
        enable_drift_compensation=bool(rng.integers(2)), min_height_drift_cnt=float(rng.choice([1, 100])),
        drift_compensation_alpha=float(rng.choice([0.1, 1.0])), overlap_clear_range_xy=float(rng.choice([1.0, 4.0])),
        overlap_clear_range_z=float(rng.choice([0.3, 2.0])), traversability_inlier=float(rng.choice([0.0, 0.1, 0.9])),
        drift_compensation_variance_inlier=float(rng.choice([0.1, 5.0])),
    )
    mode = "reference_fp16" if rng.uniform() < 0.6 else "fp32"
    N = int(rng.choice([1, 63, 1000, 20000, 45000]))
    extra = int(rng.choice([0, 0, 2]))
    R = fx.rot(*rng.uniform(-0.5, 0.5, 3))
    t = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0.5, 1.5)], np.float32)
    scatter = str(rng.choice(["auto", "atomic", "binned"]))
    return C, cfg, mode, N, extra, R, t, scatter


@pytest.mark.parametrize("seed", range(16))
def test_random_configuration(seed, weights):
    C, cfg, mode, N, extra, R, t, scatter = _random_case(seed)
    hip, orc = make_pair(cfg, C, mode, weights)
    hip.set_scatter_mode(scatter)
    rng = np.random.default_rng(1000 + seed)
    for f in range(3):
        p = fx.cloud(C, N, 10 * seed + f, dz=float(rng.uniform(-0.3, 0.1)), extra=extra)
        if N > 100:
            p[rng.integers(0, N, 5)] = np.nan
            p[: N // 20, :2] = p[0, :2]                       # a pile of points in one cell
        pn, on = float(rng.choice([0.0, 1.0])), float(rng.choice([0.0, 1.0]))
        hip.update_map_with_kernel(p, [], R, t.copy(), pn, on)
        orc.update_map_with_kernel(p, R, t, pn, on)
        for k in range(int(rng.integers(0, 8))):
            hip.update_time(); orc.update_time()
        if rng.uniform() < 0.5:
            hip.update_variance(); orc.update_variance()
        tag = "seed %d frame %d (C=%d N=%d %s %s rays=%s)" % (seed, f, C, N, mode, scatter, cfg["enable_visibility_cleanup"])
        assert_planes_close(hip.elevation_map, orc.elevation_map, what=tag)
        assert_planes_close(hip.normal_map, orc.normal_map, names=["nx", "ny", "nz"], what=tag)
    assert abs(hip.get_additive_mean_error() - float(orc.additive_mean_error)) < 1e-6
