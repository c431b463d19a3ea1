"""GPU: the asynchronous entry point (emap_upload_points behind ElevationMap.input_pointcloud, EM/elevation_mapping.py:434-466).
The caller's array may be reused the moment the call returns (the ROS wrapper's Eigen buffer is): frames fed from ONE host buffer
that is scribbled over right after every call must give the map of frames fed from private copies -- for float64 (what
src/elevation_mapping_wrapper.cpp:173-177 passes) and float32 clouds, with extra channels, over enough points that several workers
and DMA chunks are in flight, and with NaN rows (dropped, :458)."""
import numpy as np
import pytest

import _fixtures as fx
from _util import make_pair
from oracle import emap_oracle as eo

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_caller_buffer_is_free_when_the_call_returns(dtype, weights):
    C, N, F = 512, 600_000, 6
    cfg = dict(eo.YAML, enable_visibility_cleanup=False)
    R, t = fx.POSES["rotated"]
    clouds = []
    for f in range(F):
        p = np.concatenate([fx.cloud(C, N, f, dz=-0.01 * f), np.random.default_rng(f).random((N, 1), dtype=np.float32)], axis=1).astype(dtype)
        p[::997, f % 3] = np.nan                                   # NaN rows are dropped by the kernels
        clouds.append(p)
    maps = []
    for shared in (True, False):
        hip, _ = make_pair(cfg, C, "reference_fp16", weights)
        buf = np.empty_like(clouds[0])
        for f in range(F):
            if shared:
                buf[...] = clouds[f]
                hip.input_pointcloud(buf, ["x", "y", "z", "feat"], R, t.copy() + hip.center, 1.0, 1.0)
                buf[...] = np.nan if f % 2 else 1e6                 # the caller reuses its buffer immediately
            else:
                hip.input_pointcloud(clouds[f].copy(), ["x", "y", "z", "feat"], R, t.copy() + hip.center, 1.0, 1.0)
        maps.append(hip.elevation_map.tobytes() + hip.normal_map.tobytes() + hip.semantic_map.semantic_map.tobytes())
        assert int((hip.elevation_map[2] > 0.5).sum()) > 0.5 * C * C
        hip.close()
    assert maps[0] == maps[1]


def test_upload_matches_the_oracle_on_a_float64_cloud(weights):
    C, N = 202, 150_000
    hip, orc = make_pair(eo.YAML, C, "reference_fp16", weights)
    R, t = fx.POSES["identity"]
    for f in range(3):
        p = fx.cloud(C, N, 30 + f, dz=-0.05 * f)
        hip.input_pointcloud(p.astype(np.float64), ["x", "y", "z"], R, t.copy() + hip.center, 0.0, 0.0)
        orc.update_map_with_kernel(p, R, t, 0.0, 0.0)
    from _util import assert_planes_close
    assert_planes_close(hip.elevation_map, orc.elevation_map, what="float64 upload")
