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


@pytest.mark.parametrize("dist", [False, True])
def test_image_factories(dist, weights):
    """EM/kernels/custom_image_kernels.py through the factories: the correspondence kernel (projection, radtan, Bresenham occlusion
    walk) and the three samplers -- exact against the oracle, which equals the reference's own kernel source
    (tests/test_oracle_vs_reference_source.py::test_image_correspondence_and_fusions)."""
    from elevation_mapping_cupy import kernels as K
    C = 98
    cfg = dict(eo.YAML, enable_visibility_cleanup=False)
    om = eo.OracleMap(eo.make_params(cfg, cell_n=C, weights=weights))
    R0, t0 = fx.POSES["identity"]
    p = fx.cloud(C, 20000, 0); p[:, 2] += 0.3 * np.sin(p[:, 0] * 2.0)           # relief => occlusions
    om.update_map_with_kernel(p, R0, t0)
    Kc, D, R, t, H, W = fx.camera_case(C, 1, dist)
    center = np.array([0.1, -0.2, 0.05], np.float32)
    Pm, x1, y1, z1 = fx.camera_inputs(center, C, 0.04, Kc, R, t)
    uv_o, va_o = eo.image_correspondence(om.P, om.elevation_map, x1, y1, z1, Pm.ravel(), Kc.ravel(), D, H, W, center)
    uv = np.zeros((2, C, C), np.float32); va = np.zeros((C, C), np.bool_)
    K.image_to_map_correspondence_kernel(0.04, C, C, 0.10)(om.elevation_map.copy(), x1, y1, z1, Pm.ravel().copy(), Kc.ravel().copy(), D.copy(),
                                                             H, W, center, uv, va, size=C * C)
    assert va.sum() > 50 and np.array_equal(va, va_o.astype(bool)) and np.array_equal(uv, uv_o)
    # a tighter collision tolerance rejects more cells (the factory parameter reaches the kernel)
    va2 = np.zeros((C, C), np.bool_)
    K.image_to_map_correspondence_kernel(0.04, C, C, -0.05)(om.elevation_map.copy(), x1, y1, z1, Pm.ravel().copy(), Kc.ravel().copy(), D.copy(),
                                                              H, W, center, np.zeros_like(uv), va2, size=C * C)
    assert va2.sum() < va.sum() and not (va2 & ~va).any()
    rng = np.random.default_rng(5)
    img = rng.uniform(0, 1, (3, H, W)).astype(np.float32)
    rgb = rng.integers(0, 256, (3, H, W)).astype(np.float32)
    sem = rng.uniform(0, 1, (3, C, C)).astype(np.float32)
    new = np.zeros_like(sem)
    K.exponential_correspondences_to_map_kernel(C, C, 0.7)(sem, 0, img[1], uv, va, H, W, new, size=C * C)
    K.color_correspondences_to_map_kernel(C, C)(sem, 1, rgb, uv, va, H, W, new, size=C * C)
    K.average_correspondences_to_map_kernel(C, C)(sem, 2, img[2], uv, va, H, W, new, size=C * C)
    want = sem.copy()
    eo.image_fuse(om.P, "exponential", want[0], img[1], uv_o, va_o, H, W, 0.7)
    eo.image_fuse(om.P, "color", want[1], rgb, uv_o, va_o, H, W)
    eo.image_fuse(om.P, "average", want[2], img[2], uv_o, va_o, H, W)
    assert np.array_equal(new.view(np.uint32), want.view(np.uint32))
    assert (new[2] != sem[2]).sum() == va.sum()
