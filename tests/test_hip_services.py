"""GPU: the low-rate services of the node-facing API: safety polygon (polygon_mask_kernel vs the compiled
reference kernel, bit exact), per-layer accessors, map initialisation (dilation passes vs the oracle's dilation)."""
import numpy as np
import pytest

import _fixtures as fx
from _util import make_pair
from oracle import build_ref, emap_oracle as eo, ref_kernels

pytestmark = pytest.mark.gpu

POLYGONS = [
    np.array([[-0.8, -0.5], [0.9, -0.7], [1.1, 0.6], [-0.2, 1.2]], np.float32),                       # convex quad
    np.array([[-1.5, -1.5], [1.5, -1.5], [1.5, 1.5], [0.0, 0.2], [-1.5, 1.5]], np.float32),            # concave
    np.array([[0.3, 0.3], [0.31, 0.3], [0.3, 0.31]], np.float32),                                      # sub-cell triangle
    np.array([[-4.0, -4.0], [4.0, -4.0], [4.0, 4.0], [-4.0, 4.0]], np.float32),                        # larger than the map (clipped by the caller)
]


@pytest.mark.parametrize("k", range(len(POLYGONS)))
@pytest.mark.parametrize("center", [(0.0, 0.0), (0.52, -0.28)])
def test_polygon_mask_vs_reference_kernel(k, center, weights):
    params = build_ref.PREBUILD["polygon130"]
    if not ref_kernels.available(params):
        pytest.skip("compiled reference not built")
    rk = ref_kernels.RefKernels(params, build=False)
    C = 130
    hip, _ = make_pair(eo.DEFAULTS, C, "reference_fp16", weights)
    hip.center[:2] = center
    L = C * 0.04
    poly = POLYGONS[k] + np.array(center, np.float32)
    pmin = np.array(center) - (L - 0.08) / 2 + 0.04; pmax = np.array(center) + (L - 0.08) / 2 - 0.04
    poly = np.clip(poly, pmin, pmax).astype(np.float32)
    got = hip.polygon_mask(poly)
    want = np.full((C, C), -1, np.float32)
    bbox = np.concatenate([poly.min(axis=0), poly.max(axis=0)]).astype(np.float32)
    rk.polygon_mask(poly, center[0], center[1], bbox, want)
    assert np.array_equal(got, want)
    assert k == 2 or got.sum() > 10


def test_polygon_traversability_service(weights):
    C = 130
    hip, _ = make_pair(eo.DEFAULTS, C, "reference_fp16", weights)
    R, t = fx.POSES["identity"]
    hip.update_map_with_kernel(fx.cloud(C, 60000, 0), [], R, t.copy(), 0.0, 0.0)
    poly = np.array([[-1.0, -1.0], [1.0, -1.0], [1.0, 1.0], [-1.0, 1.0]], np.float64)
    result = np.zeros(3)
    n = hip.get_polygon_traversability(poly, result)
    assert abs(result[2] - 4.0) < 1e-9 and 0.0 <= result[1] <= 1.0 and result[0] in (0.0, 1.0)
    # cross-check the mean untraversability with plain numpy on the downloaded layers
    m = hip.elevation_map
    mask = hip.mask[1:-1, 1:-1]
    un = np.where(m[2][1:-1, 1:-1] > 0.5, 1 - m[3][1:-1, 1:-1], 0) * mask
    assert abs(result[1] - un.sum() / (m[2][1:-1, 1:-1] * mask).sum()) < 1e-6
    if n:
        out = np.zeros((n, 2), np.float32)
        hip.get_untraversable_polygon(out)
        assert np.allclose(out[0], out[-1]) and np.abs(out).max() <= 1.0 + 0.04 * 2        # closed ring inside the polygon
    # a polygon outside the map is unsafe
    far = poly + 100.0
    hip.get_polygon_traversability(far, result)
    assert result[0] == 0.0


def test_layer_accessors_match_get_map_with_name_ref(weights):
    C = 98
    hip, _ = make_pair(eo.YAML, C, "reference_fp16", weights)
    R, t = fx.POSES["rotated"]
    hip.update_map_with_kernel(fx.cloud(C, 30000, 1), [], R, t.copy(), 0.0, 0.0)
    hip.center[2] = 0.37
    for name, fn in [("elevation", hip.get_elevation), ("variance", hip.get_variance), ("traversability", hip.get_traversability),
                     ("time", hip.get_time), ("upper_bound", hip.get_upper_bound), ("is_upper_bound", hip.get_is_upper_bound)]:
        ref = np.zeros((C - 2, C - 2), np.float32)
        hip.get_map_with_name_ref(name, ref)
        got = np.asarray(fn(), np.float32)
        assert got.shape == (C - 2, C - 2)
        assert np.array_equal(np.flip(got), ref, equal_nan=True), name
    nx, ny, nz = (np.zeros((C - 2, C - 2), np.float32) for _ in range(3))
    hip.get_normal_ref(nx, ny, nz)
    assert np.array_equal(hip.get_normal_maps(), np.stack([nx, ny, nz]))
    assert np.array_equal(hip.process_map_for_publish(hip.elevation_map[1]), hip.get_variance())


def test_initialize_map(weights):
    C = 66
    hip, _ = make_pair(dict(eo.DEFAULTS, dilation_size_initialize=3), C, "reference_fp16", weights)
    hip.center[:] = [0.4, -0.2, 0.5]
    pts = np.array([[-0.6, -0.9, 0.45], [0.9, -0.8, 0.55], [1.2, 0.6, 0.65], [-0.3, 0.7, 0.5]], np.float64)
    hip.initialize_map(pts, method="linear")
    m = hip.elevation_map
    assert (m[2] > 0.5).sum() > 400
    valid = m[2] > 0.5
    assert np.all(m[0][valid] >= -0.06) and np.all(m[0][valid] <= 0.16)          # heights relative to center z, inside the hull's range
    assert np.array_equal(m[5][valid], m[0][valid]) and not m[6][valid].any()
    # the two dilation passes grow the interpolated region exactly like the oracle's dilation applied twice (out of place)
    from scipy.interpolate import griddata
    idx = ((pts[:, :2].astype(np.float32) - hip.center[:2].astype(np.float32).reshape(1, 2)) / 0.04 + C / 2).astype(np.int32)
    gx, gy = np.mgrid[0:C, 0:C]
    z = (pts[:, 2].astype(np.float32) - np.float32(0.5))
    interp = griddata(idx.astype(np.float32), z, (gx, gy), method="linear")
    e0 = np.nan_to_num(interp).astype(np.float32); v0 = (~np.isnan(interp)).astype(np.float32)
    e1, v1 = eo.dilate_plane(C, 3, e0, v0); v1 = np.where(v0 > 0.5, v0, v1)
    e2, v2 = eo.dilate_plane(C, 3, e1, v1); v2 = np.where(v1 > 0.5, v1, v2)
    assert np.array_equal(m[2], v2) and np.allclose(m[0], e2, atol=1e-6)
    assert np.allclose(m[1][v0 > 0.5], 10.0) and np.allclose(m[1][v0 < 0.5], hip.initial_variance)


def test_move_relative_shift_sign_convention(weights):
    """move(delta) rolls by +delta_pixel (reference :139-152) whereas move_to rolls by -delta_pixel (:154-170)"""
    C = 66
    hip, _ = make_pair(eo.DEFAULTS, C, "reference_fp16", weights)
    R, t = fx.POSES["identity"]
    hip.update_map_with_kernel(fx.cloud(C, 6000, 0), [], R, t.copy(), 0.0, 0.0)
    before = hip.elevation_map
    hip.move(np.array([2 * 0.04, -1 * 0.04, 0.1], np.float32))
    after = hip.elevation_map
    want = np.roll(before, (2, -1), axis=(1, 2)); want[:, :2, :] = 0; want[:, :, -1:] = 0
    want[1, :2, :] = hip.initial_variance; want[1, :, -1:] = hip.initial_variance
    want[0] -= np.float32(0.1); want[5] -= np.float32(0.1)
    assert np.allclose(after, want, atol=1e-6)
    assert np.allclose(hip.center, [0.08, -0.04, 0.1], atol=1e-6)
