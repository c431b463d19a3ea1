"""GPU: parity at BASELINE.json's FULL sizes (configs 2 and 3: 1024 x 1024 map, 1 M points per frame), against the C oracle run
with its OpenMP mode (seconds per frame) and through size-independent properties of the path:
  * cell indices / valid / inside flags of all 1 M points bit-exact,
  * every plane within 1e-5 after warm frames with the drift gate open (config 2) and with rays + overlap clearance (config 3),
  * permutation invariance: the fused map does not depend on the order of the points (integer accumulators) -- bit for bit,
  * conservation: on a fresh map every valid & inside point is accepted, so #valid cells == #distinct cell indices."""
import numpy as np
import pytest

import _fixtures as fx
from _util import assert_planes_close, assert_planes_equal, make_pair
from oracle import emap_oracle as eo

pytestmark = pytest.mark.gpu
C, N = 1024, 1_000_000


@pytest.fixture(autouse=True)
def _oracle_threads():
    eo.set_threads(8)
    yield
    eo.set_threads(1)


def test_config2_full_size_vs_oracle(weights):
    cfg = dict(eo.YAML, enable_visibility_cleanup=False, enable_overlap_clearance=False)
    hip, orc = make_pair(cfg, C, "reference_fp16", weights)
    R, t = fx.POSES["rotated"]
    p0 = fx.cloud(C, N, 0)
    hip.bind_points(p0)
    idx, valid, inside = hip.point_index(R, t)
    o_idx, o_valid, o_inside = orc.point_index(p0, R, t)
    assert np.array_equal(valid, o_valid) and np.array_equal(inside, o_inside)
    use = (valid > 0) & (inside > 0)
    assert np.array_equal(idx[use], o_idx[use]) and use.sum() > 0.3 * N
    for f, dz in enumerate((0.0, -0.02, -0.06)):
        p = fx.cloud(C, N, f, dz=dz)
        hip.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0)
        orc.update_map_with_kernel(p, R, t, 1.0, 1.0)
        if f == 0:   # fresh map: no outliers, every valid & inside point is accepted
            assert int((hip.elevation_map[2] > 0.5).sum()) == np.unique(idx[use]).size
        for k in range(4):
            hip.update_time(); orc.update_time()
        hip.update_variance(); orc.update_variance()
    assert_planes_close(hip.elevation_map, orc.elevation_map, what="config 2, 3 frames")          # north_star's tolerance ...
    assert_planes_equal(hip.elevation_map, orc.elevation_map, what="config 2, 3 frames")          # ... and what actually holds: bit for bit
    assert_planes_equal(hip.normal_map, orc.normal_map, names=["nx", "ny", "nz"])
    assert hip.get_additive_mean_error() == float(orc.additive_mean_error)


def test_config3_full_size_vs_oracle(weights):
    hip, orc = make_pair(eo.YAML, C, "reference_fp16", weights)            # rays + overlap clearance on
    R, t = fx.POSES["identity"]
    for f, dz in enumerate((0.0, -0.2)):
        p = fx.cloud(C, N, f, dz=dz)
        hip.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0)
        orc.update_map_with_kernel(p, R, t, 1.0, 1.0)
        for k in range(10):
            hip.update_time(); orc.update_time()
    assert orc.last["ray_visits"] > 1e8                                       # the ray pass really ran at full size
    assert_planes_close(hip.elevation_map, orc.elevation_map, what="config 3, 2 frames")
    assert_planes_equal(hip.elevation_map, orc.elevation_map, what="config 3, 2 frames")
    assert_planes_equal(hip.normal_map, orc.normal_map, names=["nx", "ny", "nz"])


@pytest.mark.parametrize("rays", [False, True])
def test_point_order_does_not_matter_full_size(rays, weights):
    """planes 0-2, 4, 6 (elevation, variance, validity, time, upper-bound flag) are sums / minima over the point SET; plane 5 keeps
    the height of the point with the largest index by definition (what sequential execution of the reference yields)."""
    cfg = dict(eo.YAML, enable_visibility_cleanup=rays)
    R, t = fx.POSES["rotated"]
    p = fx.cloud(C, N, 3)
    outs = []
    for perm_seed in (None, 1):
        hip, _ = make_pair(cfg, C, "reference_fp16", weights)
        q = p if perm_seed is None else p[np.random.default_rng(perm_seed).permutation(N)]
        hip.update_map_with_kernel(q, [], R, t.copy(), 0.0, 0.0)
        m = hip.elevation_map
        outs.append(m[[0, 1, 2, 4, 6]].tobytes())
        hip.close()
    assert outs[0] == outs[1]


def test_ray_bitmap_in_lds_and_in_global_memory_agree(weights, tmp_path):
    """k_rays stages the inert bitmap in LDS when it fits (1024^2: 128 KB); EMAP_RAY_LMAP=0 keeps it in global memory (the path of
    larger maps).  Same frame, two processes (the switch is read once per process): identical maps, bit for bit."""
    import hashlib
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, hashlib, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import _fixtures as fx\n"
        "from _util import make_pair\n"
        "from oracle import emap_oracle as eo\n"
        "w = np.load(%r); w = {k: w[k] for k in ('w1', 'w2', 'w3', 'w_out')}\n"
        "hip, _ = make_pair(eo.YAML, 1024, 'reference_fp16', w)\n"
        "R, t = fx.POSES['rotated']\n"
        "for f in range(2):\n"
        "    hip.update_map_with_kernel(fx.cloud(1024, 400000, 7 + f, dz=-0.1 * f), [], R, t.copy(), 0.0, 0.0)\n"
        "    hip.update_time()\n"
        "print(hashlib.sha1(hip.elevation_map.tobytes() + hip.normal_map.tobytes()).hexdigest())\n"
    ) % (root, os.path.join(root, "tests"), os.path.join(root, "tests", "golden", "weights.npz"))
    digests = []
    for lmap in ("1", "0"):
        env = dict(os.environ, EMAP_RAY_LMAP=lmap)
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=root)
        assert out.returncode == 0, out.stderr[-2000:]
        digests.append(out.stdout.strip().splitlines()[-1])
    assert digests[0] == digests[1]
