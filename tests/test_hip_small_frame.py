"""GPU: k_small_frame -- the robot-scale frame (small clouds on maps of up to 512^2 cells: what the reference ships, EM/parameter.py:137,165)
runs count -> gate -> fuse -> commit / average as ONE launch with two grid-wide barriers (emap_kernels.hip).  The same frames through the
staged API are a chain of launches of the older kernels (k_count, k_gate, k_fuse, k_commit, k_average): every plane must agree BIT FOR
BIT between the two (integer / fixed-point accumulators, the same float statements), and with the oracle (reference:
EM/elevation_mapping.py:316-391, EM/kernels/custom_kernels.py:160-197,280-389)."""
import numpy as np
import pytest

import _fixtures as fx
from _util import assert_planes_equal, make_pair
from oracle import emap_oracle as eo

pytestmark = pytest.mark.gpu
NORMALS = ["nx", "ny", "nz"]


def _staged(hip, p, R, t, pn, on):
    """one frame as separate launches (the staged API never takes the one-launch path)"""
    P = hip.param
    hip.bind_points(p)
    hip.stage("count", R, t)
    hip.stage("gate", position_noise=pn, orientation_noise=on)
    hip.stage("fuse", R, t)
    hip.stage("commit")
    if P.enable_visibility_cleanup:
        hip.stage("rays", R, t)
    hip.stage("average")
    if P.enable_overlap_clearance:
        hip.stage("overlap", t=float(np.float32(t[2])))
    hip.stage("post")


def _same(a, b, what):
    assert_planes_equal(a.elevation_map, b.elevation_map, what=what)
    assert_planes_equal(a.normal_map, b.normal_map, names=NORMALS, what=what + " normals")
    assert a.traversability_input.tobytes() == b.traversability_input.tobytes(), what + " traversability_input"


@pytest.mark.parametrize("mode", ["reference_fp16", "fp32"])
@pytest.mark.parametrize("rays", [False, True])
@pytest.mark.parametrize("cfg_name,C,N", [("yaml", 202, 50000), ("default", 202, 20000), ("yaml", 130, 9000), ("yaml", 512, 131071)])
def test_one_launch_equals_the_chain_and_the_oracle(cfg_name, C, N, rays, mode, weights):
    cfg = dict(eo.YAML if cfg_name == "yaml" else eo.DEFAULTS, enable_visibility_cleanup=rays)
    one, orc = make_pair(cfg, C, mode, weights)
    chain, _ = make_pair(cfg, C, mode, weights)
    R, t = fx.POSES["rotated"]
    frames = [(0.0, 0.0, 0.0), (-0.02, 1.0, 1.0), (-0.2, 1.0, 1.0), (0.05, 1.0, 0.0)]      # (cloud offset, position noise, orientation noise): gate closed, fired, fired + outliers
    for f, (dz, pn, on) in enumerate(frames):
        p = fx.cloud(C, N, f, dz=dz)
        st = one.update_map_with_kernel(p, [], R, t.copy(), pn, on)
        # (512 workgroups -- the largest cloud of the atomic path -- need a device that holds 2048 of them: an unpartitioned MI355X)
        assert one.last_update_path() == ("small_frame" if N <= 65536 else one.last_update_path())
        _staged(chain, p, R, t, pn, on)
        orc.update_map_with_kernel(p, R, t, pn, on)
        sc = chain.stats()
        assert (st.err_cnt, st.gate_fired, st.n_points) == (sc.err_cnt, sc.gate_fired, sc.n_points), "frame %d" % f
        assert np.float32(st.shift).tobytes() == np.float32(sc.shift).tobytes() and st.err_sum == sc.err_sum, "frame %d" % f
        assert np.float32(st.additive_mean_error).tobytes() == np.float32(sc.additive_mean_error).tobytes()
        _same(one, chain, "frame %d" % f)
        for k in range(6 if f != 1 else 12):
            one.update_time(); chain.update_time(); orc.update_time()
        if f == 1:
            one.update_variance(); chain.update_variance(); orc.update_variance()
    assert_planes_equal(one.elevation_map, orc.elevation_map, what="oracle")
    assert_planes_equal(one.normal_map, orc.normal_map, names=NORMALS, what="oracle normals")


@pytest.mark.parametrize("rays", [False, True])
def test_pending_map_moves_are_replayed_inside_the_one_launch(rays, weights):
    """move -> frame: the per-cell phase writes the pending shifts out (cell_now), as k_commit / k_average do"""
    C, N = 202, 30000
    cfg = dict(eo.YAML, enable_visibility_cleanup=rays)
    one, _ = make_pair(cfg, C, "reference_fp16", weights)
    chain, _ = make_pair(cfg, C, "reference_fp16", weights)
    R, t = fx.POSES["identity"]
    for f, mv in enumerate([None, (3, -5), (-17, 2), (1, 1)]):
        if mv is not None:
            for m in (one, chain):
                m.shift_map_xy(np.array(mv)); m.shift_map_z(0.01 * f)
            if f == 3:      # two moves pending at once
                for m in (one, chain):
                    m.shift_map_xy(np.array([-2, 4]))
        p = fx.cloud(C, N, f, dz=-0.03 * f)
        one.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0)
        assert one.last_update_path() == "small_frame"
        _staged(chain, p, R, t, 1.0, 1.0)
        _same(one, chain, "frame %d" % f)
        for k in range(7):
            one.update_time(); chain.update_time()


@pytest.mark.parametrize("N", [1, 7, 300, 4097])
def test_tiny_and_ragged_clouds(N, weights):
    """a cloud smaller than the per-cell phase's grid; NaN rows; points outside the map and on its border"""
    C = 202
    cfg = dict(eo.YAML)
    one, orc = make_pair(cfg, C, "reference_fp16", weights)
    chain, _ = make_pair(cfg, C, "reference_fp16", weights)
    R, t = fx.POSES["rotated"]
    warm = fx.cloud(C, 40000, 9)
    for m in (one, chain):
        m.update_map_with_kernel(warm, [], R, t.copy(), 1.0, 1.0)
    orc.update_map_with_kernel(warm, R, t, 1.0, 1.0)
    for k in range(8):
        one.update_time(); chain.update_time(); orc.update_time()
    p = fx.cloud(C, N, 3, dz=-0.05)
    p[::3, 0] *= 1.3                       # some beyond the map
    if N > 2:
        p[1] = np.nan; p[N // 2, 2] = np.nan
    one.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0)
    assert one.last_update_path() == "small_frame"
    _staged(chain, p, R, t, 1.0, 1.0)
    orc.update_map_with_kernel(p, R, t, 1.0, 1.0)
    _same(one, chain, "N = %d" % N)
    assert_planes_equal(one.elevation_map, orc.elevation_map, what="oracle")


def test_larger_maps_and_clouds_keep_their_paths(weights):
    C = 1024
    hip, _ = make_pair(dict(eo.YAML, enable_visibility_cleanup=False), C, "reference_fp16", weights)
    R, t = fx.POSES["identity"]
    hip.update_map_with_kernel(fx.cloud(C, 20000, 0), [], R, t.copy(), 1.0, 1.0)
    assert hip.last_update_path() == "atomic"            # beyond 512^2 cells: the chain of launches
    hip.update_map_with_kernel(fx.cloud(C, 200000, 1), [], R, t.copy(), 1.0, 1.0)
    assert hip.last_update_path() == "binned"
    small, _ = make_pair(dict(eo.YAML, enable_visibility_cleanup=False), 202, "reference_fp16", weights)
    small.update_map_with_kernel(fx.cloud(202, 200000, 1), [], R, t.copy(), 1.0, 1.0)
    assert small.last_update_path() == "binned"


def test_run_to_run_and_two_contexts_bit_identical(weights):
    """40 frames on two contexts that alternate on the device: the barriers' epochs / ticket words are per context"""
    C, N = 202, 50000
    a, _ = make_pair(eo.YAML, C, "reference_fp16", weights)
    b, _ = make_pair(eo.YAML, C, "reference_fp16", weights)
    R, t = fx.POSES["rotated"]
    for f in range(40):
        p = fx.cloud(C, N, f % 5, dz=-0.01 * (f % 7))
        a.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0, want_stats=False)
        b.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0, want_stats=False)
        if f % 3 == 0:
            a.update_time(); b.update_time()
    assert a.last_update_path() == "small_frame"
    assert a.elevation_map.tobytes() == b.elevation_map.tobytes() and a.normal_map.tobytes() == b.normal_map.tobytes()

