"""GPU: k_small_frame -- the robot-scale frame (small clouds on maps of up to 512^2 cells: what the reference ships, EM/parameter.py:137,165)
without a visibility pass runs count -> gate -> fuse -> commit + average as ONE launch with two grid-wide barriers (emap_kernels.hip).
A barrier that foreign work on the GPU keeps from completing ABORTS consistently, the launch leaves everything as it found it and the
library re-runs the frame on the chain of launches (round 6): forced aborts at either barrier, pipelined frames behind an aborted one,
and frames under a foreign kernel that holds most of the compute units -- the map is bit for bit the chain's, always.  The same frames through the
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
        # (512 workgroups -- the largest cloud of the atomic path -- need a device that holds 2048 of them: an unpartitioned MI355X; a frame
        # with a visibility pass keeps the chain of launches: an aborted launch must leave nothing for the rest of the frame to act on)
        assert one.last_update_path() == ("atomic" if rays else ("small_frame" if N <= 65536 else one.last_update_path()))
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
        assert one.last_update_path() == ("atomic" if rays else "small_frame")
        _staged(chain, p, R, t, 1.0, 1.0)
        _same(one, chain, "frame %d" % f)
        for k in range(7):
            one.update_time(); chain.update_time()


@pytest.mark.parametrize("N", [1, 7, 300, 4097])
def test_tiny_and_ragged_clouds(N, weights):
    """a cloud smaller than the per-cell phase's grid; NaN rows; points outside the map and on its border"""
    C = 202
    cfg = dict(eo.YAML, enable_visibility_cleanup=False)
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
    cfg = dict(eo.YAML, enable_visibility_cleanup=False)
    a, _ = make_pair(cfg, C, "reference_fp16", weights)
    b, _ = make_pair(cfg, C, "reference_fp16", weights)
    R, t = fx.POSES["rotated"]
    for f in range(40):
        p = fx.cloud(C, N, f % 5, dz=-0.01 * (f % 7))
        a.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0, want_stats=False)
        b.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0, want_stats=False)
        if f % 3 == 0:
            a.update_time(); b.update_time()
    assert a.last_update_path() == "small_frame"
    assert a.elevation_map.tobytes() == b.elevation_map.tobytes() and a.normal_map.tobytes() == b.normal_map.tobytes()



# ---- round 6: a grid barrier that cannot complete aborts; the frame is re-run on the chain ----------------------------------------------
NO_RAYS = dict(eo.YAML, enable_visibility_cleanup=False)


class _env:
    def __init__(self, **kv): self.kv = kv
    def __enter__(self):
        import os
        for k, v in self.kv.items():
            os.environ[k] = str(v)
    def __exit__(self, *a):
        import os
        for k in self.kv:
            os.environ.pop(k, None)


@pytest.mark.parametrize("barrier", [1, 2])
@pytest.mark.parametrize("mode", ["reference_fp16", "fp32"])
def test_forced_abort_is_recovered_bit_for_bit(barrier, mode, weights):
    """workgroup 0 gives up at once at the first / second barrier of every second frame (EMAP_SF_TEST_ABORT): the launch leaves map,
    accumulators and drift record as it found them and the frame is re-run on the chain -- pending map moves included, statistics
    (err_cnt, shift, additive_mean_error) included"""
    C, N = 202, 50000
    one, _ = make_pair(NO_RAYS, C, mode, weights)
    chain, _ = make_pair(NO_RAYS, C, mode, weights)
    R, t = fx.POSES["rotated"]
    frames = [(0.0, 0.0, 0.0), (-0.02, 1.0, 1.0), (-0.2, 1.0, 1.0), (0.05, 1.0, 0.0), (-0.04, 1.0, 1.0), (0.0, 1.0, 1.0)]
    for f, (dz, pn, on) in enumerate(frames):
        if f in (2, 5):
            for m in (one, chain):
                m.shift_map_xy(np.array([3, -5])); m.shift_map_z(0.01 * f)
        p = fx.cloud(C, N, f, dz=dz)
        if f % 2 == 1:
            with _env(EMAP_SF_TEST_ABORT=barrier):
                st = one.update_map_with_kernel(p, [], R, t.copy(), pn, on)
        else:
            st = one.update_map_with_kernel(p, [], R, t.copy(), pn, on)
        _staged(chain, p, R, t, pn, on)
        sc = chain.stats()
        assert (st.err_cnt, st.gate_fired, st.n_points) == (sc.err_cnt, sc.gate_fired, sc.n_points), "frame %d" % f
        assert np.float32(st.shift).tobytes() == np.float32(sc.shift).tobytes() and st.err_sum == sc.err_sum, "frame %d" % f
        assert np.float32(st.additive_mean_error).tobytes() == np.float32(sc.additive_mean_error).tobytes(), "frame %d" % f
        _same(one, chain, "frame %d" % f)
        for k in range(6):
            one.update_time(); chain.update_time()
    assert one.small_frame_aborts() == 3, one.small_frame_aborts()      # (the chain itself == the oracle: test_one_launch_equals_the_chain_and_the_oracle; the oracle has no map moves)


@pytest.mark.parametrize("barrier", [1, 2])
def test_frames_queued_behind_an_aborted_one_are_rerun_in_order(barrier, weights):
    """device-resident clouds, no synchronisation between the frames: the launch of frame 2 aborts while frames 3 .. 6 are already
    queued -- they find the poison word, do nothing, and all five are re-run in order at the next call that looks at the map"""
    import bench
    C, N = 202, 40000
    one, _ = make_pair(NO_RAYS, C, "reference_fp16", weights)
    chain, _ = make_pair(NO_RAYS, C, "reference_fp16", weights)
    R, t = fx.POSES["rotated"]
    rt = bench.Hip()
    clouds = [fx.cloud(C, N, f, dz=-0.01 * f) for f in range(7)]
    dev = []
    for p in clouds:
        d = rt.malloc(p.nbytes); rt.h2d(d, p); dev.append(d)
    for f, p in enumerate(clouds):
        one.bind_points_device(dev[f].value, N, 3)
        if f == 2:
            with _env(EMAP_SF_TEST_ABORT=barrier):
                one.update_map_with_kernel(None, [], R, t.copy(), 1.0, 1.0, want_stats=False)
        else:
            one.update_map_with_kernel(None, [], R, t.copy(), 1.0, 1.0, want_stats=False)
        _staged(chain, p, R, t, 1.0, 1.0)
    _same(one, chain, "after the queue")           # (the first read settles the frames in flight)
    assert one.small_frame_aborts() == 5, one.small_frame_aborts()
    so, sc = one.stats(), chain.stats()
    assert np.float32(so.additive_mean_error).tobytes() == np.float32(sc.additive_mean_error).tobytes() and so.err_cnt == sc.err_cnt
    one.update_map_with_kernel(clouds[0], [], R, t.copy(), 1.0, 1.0)       # and the one-launch path goes on afterwards
    assert one.last_update_path() == "small_frame" and one.small_frame_aborts() == 5
    _staged(chain, clouds[0], R, t, 1.0, 1.0)
    _same(one, chain, "the frame after")
    one.sync()
    for d in dev:
        rt.free(d)


def test_frames_under_a_foreign_kernel_that_holds_the_compute_units(weights):
    """a foreign grid of 1024-thread workgroups (two fill a CU's wave slots) occupies all but a few CUs for 60 ms while frames are
    issued with a short patience (EMAP_SF_SPIN_LIMIT): whichever launches manage to run all their workgroups pass, the others abort
    and are re-run -- the map never differs from the chain's"""
    from _util import cu_hog
    hog = cu_hog()
    C, N = 202, 50000
    one, _ = make_pair(NO_RAYS, C, "reference_fp16", weights)
    chain, _ = make_pair(NO_RAYS, C, "reference_fp16", weights)
    R, t = fx.POSES["rotated"]
    import time
    assert hog.hog_start(0, 8, 1.0) == 0 and hog.hog_wait() == 0          # (loads the foreign code object: 0.2 s the first time)
    aborted = 0
    for rnd, groups in enumerate((500, 508, 480)):
        p0 = fx.cloud(C, N, rnd)
        one.update_map_with_kernel(p0, [], R, t.copy(), 1.0, 1.0); _staged(chain, p0, R, t, 1.0, 1.0)
        one.sync()
        assert hog.hog_start(0, groups, 60.0) == 0
        time.sleep(0.003)                                                 # the foreign grid is resident before the frames are issued
        with _env(EMAP_SF_SPIN_LIMIT=2000):
            for f in range(3):
                p = fx.cloud(C, N, 10 + 3 * rnd + f, dz=-0.02 * f)
                one.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0, want_stats=False)
                _staged(chain, p, R, t, 1.0, 1.0)
            _same(one, chain, "round %d" % rnd)
        assert hog.hog_wait() == 0
        aborted = one.small_frame_aborts()
    print("frames re-run under contention:", aborted)
    assert aborted >= 1          # (with 480 .. 508 of the 512 half-CU slots taken, part of a 196-workgroup grid cannot start: measured, every round aborts)
    p = fx.cloud(C, N, 99)
    one.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0); _staged(chain, p, R, t, 1.0, 1.0)
    assert one.last_update_path() == "small_frame"
    _same(one, chain, "afterwards")
