"""GPU: camera path through ElevationMap.input_image vs the oracle (which equals the reference kernels exactly,
tests/test_oracle_vs_reference_source.py::test_image_correspondence_and_fusions)."""
import numpy as np
import pytest

import _fixtures as fx
from _util import make_pair
from oracle import emap_oracle as eo

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("dist", [False, True])
def test_input_image_matches_oracle(dist, seed, weights):
    C = [98, 130, 66][seed % 3]
    cfg = dict(eo.YAML, enable_visibility_cleanup=False)
    hip, orc = make_pair(cfg, C, "reference_fp16", weights)
    hip.param.image_channel_fusions = {"rgb": "color", "default": "exponential"}
    R0, t0 = fx.POSES["identity"]
    p = fx.cloud(C, 20000, 0); p[:, 2] += 0.3 * np.sin(p[:, 0] * 2.0)
    hip.input_pointcloud(p, ["x", "y", "z"], R0, t0.copy(), 0.0, 0.0)
    orc.update_map_with_kernel(p, R0, t0)
    hip.move_to(np.array([0.12, -0.2, 0.05], np.float32), np.eye(3))          # non-trivial map centre
    orc.elevation_map = hip.elevation_map                                    # same shifted state
    K, D, R, t, H, W = fx.camera_case(C, seed, dist)
    rng = np.random.default_rng(5)
    feat = rng.uniform(0, 1, (H, W)).astype(np.float32)
    rgb = rng.integers(0, 256, (3, H, W)).astype(np.float32)
    # first a mono feature image, then an RGB image (the reference's colour fusion reads planes 0..2 of the stack)
    hip.input_image([feat], ["feat"], R, t, K, D, "radtan", H, W)
    Pm, x1, y1, z1 = fx.camera_inputs(hip.center, C, 0.04, K, R, t)
    uv, va = eo.image_correspondence(orc.P, orc.elevation_map, x1, y1, z1, Pm.ravel(), K.ravel(), D, H, W, hip.center)
    huv, hva = hip.get_image_correspondence()
    assert va.sum() > 50 and np.array_equal(hva, va.astype(bool)) and np.array_equal(huv, uv)
    sem = np.zeros((2, C, C), np.float32)
    eo.image_fuse(orc.P, "exponential", sem[0], feat, uv, va, H, W, 0.7)
    hip.input_image([rgb[0], rgb[1], rgb[2]], ["rgb", "g_", "b_"], R, t, K, D, "radtan", H, W)
    assert hip.semantic_map.layer_names[:2] == ["feat", "rgb"]
    eo.image_fuse(orc.P, "color", sem[1], rgb, uv, va, H, W)
    got = hip.semantic_map.semantic_map
    assert np.array_equal(got[0], sem[0])
    assert np.array_equal(got[1].view(np.uint32), sem[1].view(np.uint32))
    # second exponential frame blends with the previous value
    hip.input_image([feat * 0.5], ["feat"], R, t, K, D, "radtan", H, W)
    eo.image_fuse(orc.P, "exponential", sem[0], feat * 0.5, uv, va, H, W, 0.7)
    assert np.array_equal(hip.semantic_map.semantic_map[0], sem[0])
