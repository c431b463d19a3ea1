"""GPU: the reference's kernel-factory surface (elevation_mapping_cupy.kernels, EM/kernels/custom_kernels.py) on NumPy arrays routed to
the staged C-ABI calls -- the call pattern of ElevationMap.update_map_with_kernel (EM/elevation_mapping.py:334-391) written against the
factories reproduces the oracle's frame."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "compat"))

from oracle import emap_oracle as eo  # noqa: E402
import _fixtures as fx  # noqa: E402
from _util import assert_planes_close  # noqa: E402


def _factories(cfg, C):
    from elevation_mapping_cupy import kernels as K
    r = cfg["resolution"]
    ec = K.error_counting_kernel(r, C, C, cfg["sensor_noise_factor"], cfg["mahalanobis_thresh"], cfg["drift_compensation_variance_inlier"],
                                 cfg["traversability_inlier"], cfg["min_valid_distance"], cfg["max_height_range"],
                                 cfg["ramped_height_range_a"], cfg["ramped_height_range_b"], cfg["ramped_height_range_c"])
    ap = K.add_points_kernel(r, C, C, cfg["sensor_noise_factor"], cfg["mahalanobis_thresh"], cfg["outlier_variance"], cfg["wall_num_thresh"],
                             cfg["max_ray_length"], cfg["cleanup_step"], cfg["min_valid_distance"], cfg["max_height_range"],
                             cfg["cleanup_cos_thresh"], cfg["ramped_height_range_a"], cfg["ramped_height_range_b"],
                             cfg["ramped_height_range_c"], cfg["enable_edge_sharpen"], cfg["enable_visibility_cleanup"])
    av = K.average_map_kernel(C, C, cfg["max_variance"], cfg["initial_variance"])
    return ec, ap, av, K


@pytest.mark.parametrize("rays", [False, True])
def test_update_written_against_the_factories(weights, rays):
    C = 66
    cfg = dict(eo.DEFAULTS, enable_visibility_cleanup=rays, enable_overlap_clearance=False)
    P = eo.make_params(cfg, cell_n=C, mode="reference_fp16", weights=weights)
    orc = eo.OracleMap(P)
    R, t = fx.POSES["rotated"]
    ec, ap, av, K = _factories(cfg, C)
    emap = orc.elevation_map.copy()
    norm = orc.normal_map.copy()
    new_map = np.zeros_like(emap)
    zero = np.zeros(1, np.float32)
    for f in range(3):
        pts = fx.cloud(C, 6000, 20 + f)
        # EM/elevation_mapping.py:334-375 with the factories (drift compensation gate cannot fire: both noises 0)
        new_map *= 0.0
        error, error_cnt = np.zeros(1, np.float32), np.zeros(1, np.float32)
        ec(emap, pts, zero, zero, R, t, new_map, error, error_cnt, size=pts.shape[0])
        ap(zero, zero, R, t, norm, pts, emap, new_map, size=pts.shape[0])
        av(new_map, emap, size=C * C)
        _, _, s, c = orc.count(pts, R, t)
        orc.gate(0.0, 0.0); orc.fuse(pts, R, t); orc.commit()
        if rays:
            orc.rays(pts, R, t)
        orc.average()
        assert int(error_cnt[0]) == int(c) and abs(float(error[0]) - float(s)) <= 1e-3 * max(1.0, abs(float(s)))
        assert_planes_close(emap, orc.elevation_map, what="frame %d" % f)


def test_stencil_and_mask_factories(weights):
    C = 66
    cfg = dict(eo.DEFAULTS)
    P = eo.make_params(cfg, cell_n=C, mode="reference_fp16", weights=weights)
    orc = eo.OracleMap(P)
    R, t = fx.POSES["rotated"]
    orc.update_map_with_kernel(fx.cloud(C, 9000, 3), R, t.copy(), 0.0, 0.0)
    from elevation_mapping_cupy import kernels as K
    # dilation (EM/elevation_mapping.py:376-383): upper-bound plane with the combined mask
    e = orc.elevation_map
    src, mask = e[5].copy(), (e[2] + e[6]).astype(np.float32)
    out, omask = np.zeros_like(src), np.zeros_like(mask)
    K.dilation_filter_kernel(C, C, int(cfg["dilation_size"]))(src, mask, out, omask, size=C * C)
    assert np.array_equal(out, orc.traversability_input)
    # normals (:389-391) from the dilated plane
    nm = np.zeros((3, C, C), np.float32)
    K.normal_filter_kernel(C, C, cfg["resolution"])(orc.traversability_input, e[2].copy(), nm, size=C * C)
    assert_planes_close(nm, orc.normal_map, names=["nx", "ny", "nz"])
    # polygon mask (:837-889)
    poly = np.array([[-0.5, -0.4], [0.6, -0.3], [0.4, 0.7], [-0.3, 0.5]], np.float32)
    m = np.zeros((C, C), np.float32)
    K.polygon_mask_kernel(C, C, cfg["resolution"])(poly, np.zeros(1, np.float32), np.zeros(1, np.float32), np.array([4], np.int16),
                                                      np.zeros(4, np.float32), m, size=C * C)
    assert 0 < m.sum() < C * C and set(np.unique(m)) <= {0.0, 1.0}
    assert callable(K.sum_kernel(0.9, 4, 4))            # the semantic factories: tests/test_hip_semantic_factories.py
