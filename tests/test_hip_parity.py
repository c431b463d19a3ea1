"""GPU parity tests proper: every call goes through the C ABI (libemap_hip.so) and is compared with the CPU
oracle (oracle/emap_oracle.c, itself pinned to the reference's own kernel source -- tests/test_oracle_*.py)
on identical seeded inputs.  Bars (BASELINE.json north_star): cell indices / flags / integer counters bit-exact;
fused height, variance and every other float plane within 1e-5."""
import numpy as np
import pytest

import _fixtures as fx
from _util import assert_planes_close, make_pair
from oracle import emap_oracle as eo

pytestmark = pytest.mark.gpu

CASES = [  # (name, cfg, C, N)
    ("yaml202", eo.YAML, 202, 50000),          # BASELINE config 1
    ("default202", eo.DEFAULTS, 202, 50000),
    ("yaml1024", eo.YAML, 1024, 300000),       # BASELINE config 2/3 map size (oracle-sized N)
]


@pytest.mark.parametrize("mode", ["reference_fp16", "fp32"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("pose", ["identity", "rotated"])
def test_point_index_bit_exact(case, mode, pose):
    _, cfg, C, N = case
    hip, orc = make_pair(cfg, C, mode)
    R, t = fx.POSES[pose]
    p = fx.cloud(C, N, 0)
    p[::997, 1] = np.nan                                  # NaN rows are skipped (elevation_mapping.py:458)
    p[5::1009] *= 3.0                                     # points outside the map / clamped
    hip.bind_points(p)
    i1, v1, s1 = hip.point_index(R, t)
    i0, v0, s0 = orc.point_index(p, R, t)
    assert np.array_equal(i1, i0) and np.array_equal(v1, v0) and np.array_equal(s1, s0)


def _run_frame_stages(hip, orc, p, R, t, pn=0.0, on=0.0, check=True, tag=""):
    """Stage by stage, comparing every intermediate the contract defines."""
    P = orc.P
    hip.bind_points(p)
    hip.stage("count", R, t); orc.count(p, R, t)
    hip.stage("gate", position_noise=pn, orientation_noise=on); orc.gate(pn, on)
    st = hip.stats()
    assert st.err_cnt == orc.last["err_cnt"], tag
    assert abs(st.err_sum - orc.last["err_sum"]) <= 1e-6 * max(1.0, orc.last["err_cnt"]), tag
    assert bool(st.gate_fired) == orc.last["gate_fired"], tag
    assert abs(st.shift - orc.last["shift"]) <= 1e-7, tag
    hip.stage("fuse", R, t); orc.fuse(p, R, t)
    hip.stage("commit"); orc.commit()
    if check:
        assert_planes_close(hip.elevation_map, orc.elevation_map, what=tag + " S1")
    if P.enable_visibility_cleanup:
        hip.stage("rays", R, t); orc.rays(p, R, t)
    hip.stage("average"); orc.average()
    if check:
        assert_planes_close(hip.elevation_map, orc.elevation_map, what=tag + " average")
    if P.enable_overlap_clearance:
        tz = float(np.float32(t[2]))
        hip.stage("overlap", t=tz); orc.overlap_clear(tz)
    hip.stage("dilate"); orc.dilate()
    if check:
        assert np.array_equal(hip.traversability_input, orc.traversability_input), tag + " dilation must be exact"
    hip.stage("traversability_normals"); orc.traversability(); orc.normals()
    if check:
        assert_planes_close(hip.elevation_map, orc.elevation_map, what=tag + " frame end")
        assert_planes_close(hip.normal_map, orc.normal_map, names=["nx", "ny", "nz"], what=tag + " normals")


@pytest.mark.parametrize("mode", ["reference_fp16", "fp32"])
@pytest.mark.parametrize("cfg_name,C,N", [("yaml", 202, 50000), ("default", 202, 50000), ("yaml", 130, 20000)])
def test_three_frames_stagewise(cfg_name, C, N, mode, weights):
    """fresh frame, then two warm frames (time advanced, cloud lowered => outliers, drift gate, ray hits)."""
    cfg = eo.YAML if cfg_name == "yaml" else eo.DEFAULTS
    hip, orc = make_pair(cfg, C, mode, weights)
    R, t = fx.POSES["rotated"]
    _run_frame_stages(hip, orc, fx.cloud(C, N, 0), R, t, tag="f0")
    for k in range(12):
        hip.update_time(); orc.update_time()
    hip.update_variance(); orc.update_variance()
    assert_planes_close(hip.elevation_map, orc.elevation_map, what="after time/variance")
    _run_frame_stages(hip, orc, fx.cloud(C, N, 1, dz=-0.02), R, t, pn=1.0, on=1.0, tag="f1")
    for k in range(6):
        hip.update_time(); orc.update_time()
    _run_frame_stages(hip, orc, fx.cloud(C, N, 2, dz=-0.2), R, t, pn=1.0, on=1.0, tag="f2")
    assert abs(hip.get_additive_mean_error() - float(orc.additive_mean_error)) < 1e-6


@pytest.mark.parametrize("mode", ["reference_fp16", "fp32"])
def test_whole_frame_api_matches_stagewise_oracle(mode, weights):
    """emap_update (one call) == the oracle's update_map_with_kernel over 3 frames, 1024^2 map."""
    C, N = 1024, 200000
    hip, orc = make_pair(eo.YAML, C, mode, weights)
    R, t = fx.POSES["identity"]
    for f, dz in enumerate((0.0, -0.02, -0.2)):
        p = fx.cloud(C, N, f, dz=dz)
        hip.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0)
        orc.update_map_with_kernel(p, R, t, 1.0, 1.0)
        for k in range(7):
            hip.update_time(); orc.update_time()
        hip.update_variance(); orc.update_variance()
    assert_planes_close(hip.elevation_map, orc.elevation_map, what="3 frames")
    assert_planes_close(hip.normal_map, orc.normal_map, names=["nx", "ny", "nz"])


def test_run_to_run_bit_identical(weights):
    """integer / fixed-point accumulators => the same bytes every run (the reference's float atomics are not)."""
    C, N = 202, 50000
    outs = []
    for rep in range(3):
        hip, _ = make_pair(eo.YAML, C, "reference_fp16", weights)
        R, t = fx.POSES["rotated"]
        for f, dz in enumerate((0.0, -0.05)):
            hip.update_map_with_kernel(fx.cloud(C, N, f, dz=dz), [], R, t.copy(), 1.0, 1.0)
            for k in range(6):
                hip.update_time()
        outs.append(hip.elevation_map.tobytes() + hip.normal_map.tobytes())
        hip.close()
    assert outs[0] == outs[1] == outs[2]


def test_golden_frame66_against_reference_source(weights):
    """HIP vs the committed outputs of the reference's own kernels (tests/golden/frame_yaml66.npz), fresh map."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "frame_yaml66.npz"))
    C, N = 66, 6000
    for pose, (R, t) in fx.POSES.items():
        hip, _ = make_pair(eo.YAML, C, "reference_fp16", weights)
        hip.param.enable_overlap_clearance = False
        hip.reload_params()
        p = fx.cloud(C, N, 0)
        hip.bind_points(p)
        idx, valid, inside = hip.point_index(R, t)
        assert np.array_equal(idx, g[pose + "_idx"])
        assert np.array_equal(valid | (inside << 1), g[pose + "_flags"])
        hip.update_map_with_kernel(None, [], R, t.copy(), 0.0, 0.0)
        m = hip.elevation_map
        gm = g[pose + "_map"]
        for k in (0, 1, 2, 4, 5, 6):      # plane 3 is written by the traversability filter (not in the golden run)
            assert_planes_close(m[k:k + 1], gm[k:k + 1], names=[str(k)], what="golden " + pose)
        assert np.array_equal(hip.traversability_input, g[pose + "_dil"])
        assert_planes_close(hip.normal_map, g[pose + "_normal"], names=["nx", "ny", "nz"])


@pytest.mark.parametrize("d", [1, 2, 3, 10])
def test_dilation_radii_and_wraparound(d):
    """LDS-tiled dilation incl. the reference's flat-index row wrap (custom_kernels.py:403-407) at radius 1..10."""
    C = 130
    cfg = dict(eo.DEFAULTS, dilation_size=d)
    hip, orc = make_pair(cfg, C)
    rng = np.random.default_rng(d)
    e = np.zeros((7, C, C), np.float32)
    e[5] = rng.uniform(-1, 1, (C, C)); e[2] = rng.uniform(0, 1, (C, C)) < 0.05; e[6] = rng.uniform(0, 1, (C, C)) < 0.03
    e[2][:, :4] = rng.uniform(0, 1, (C, 4)) < 0.5; e[2][:, -4:] = rng.uniform(0, 1, (C, 4)) < 0.5   # busy edge columns
    hip.elevation_map = e; orc.elevation_map[...] = e
    hip.stage("dilate"); orc.dilate()
    assert np.array_equal(hip.traversability_input, orc.traversability_input)


def test_fails_loudly_without_points():
    from elevation_mapping_cupy_amd._lib import EmapError
    hip, _ = make_pair(eo.DEFAULTS, 66)
    R, t = fx.POSES["identity"]
    hip.bind_points(np.zeros((0, 3), np.float32))
    hip.update_map_with_kernel(None, [], R, t.copy(), 0, 0)        # empty cloud is legal
    with pytest.raises(EmapError):
        hip.stage("rays", R, t)                                     # rays without commit is a contract violation


def test_ragged_and_degenerate_inputs(weights):
    C = 66
    hip, orc = make_pair(eo.YAML, C, "reference_fp16", weights)
    R, t = fx.POSES["identity"]
    p = fx.cloud(C, 777, 3, extra=2)                 # N not a multiple of 64/256, extra channels (stride 5)
    p[10] = np.nan; p[11, 0] = np.inf; p[12, :3] = 0  # NaN row, inf coordinate, point at the sensor
    p[100:400, :2] = p[100, :2]                       # 300 points into one cell (> wall_num_thresh)
    p[12, :3] = [0.0, 0.0, -1.0]
    _run_frame_stages(hip, orc, p, R, t, tag="ragged f0")
    _run_frame_stages(hip, orc, p, R, t, pn=1, on=1, tag="ragged f1")


@pytest.mark.parametrize("mode", ["reference_fp16", "fp32"])
@pytest.mark.parametrize("C,N", [(202, 50000), (1024, 400000), (130, 7000)])
def test_binned_scatter_is_bit_identical_to_atomic_scatter(C, N, mode, weights):
    """tile-binned LDS reduction (emap_binned.hip) vs global-atomic scatter: same bytes in every plane, 3 frames with
    outliers, walls (> wall_num_thresh points in a cell), drift gate and rays."""
    outs = []
    for scatter in ("atomic", "binned"):
        hip, _ = make_pair(eo.YAML, C, mode, weights)
        hip.set_scatter_mode(scatter)
        R, t = fx.POSES["rotated"]
        for f, dz in enumerate((0.0, -0.02, -0.2)):
            p = fx.cloud(C, N, f, dz=dz)
            p[100:900, :2] = p[100, :2]                  # 800 points into one cell
            p[::501, 0] = np.nan
            hip.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0)
            for k in range(6):
                hip.update_time()
        outs.append((hip.elevation_map.tobytes(), hip.normal_map.tobytes(), hip.get_additive_mean_error()))
        hip.close()
    assert outs[0] == outs[1]


@pytest.mark.parametrize("mode", ["reference_fp16", "fp32"])
def test_fused_tile_kernel_without_rays(mode, weights):
    """rays off + large cloud: emap_update takes the one-kernel fuse+commit+average tile path; it must equal both the
    oracle and the atomic scatter path (bytes), incl. semantic counts."""
    C, N = 202, 60000
    cfg = dict(eo.YAML, enable_visibility_cleanup=False)
    res = {}
    for scatter in ("binned", "atomic"):
        hip, orc = make_pair(cfg, C, mode, weights)
        hip.set_scatter_mode(scatter)
        R, t = fx.POSES["rotated"]
        for f, dz in enumerate((0.0, -0.02, -0.2)):
            p = fx.cloud(C, N, f, dz=dz, extra=1)
            p[200:700, :2] = p[200, :2]
            hip.param.pointcloud_channel_fusions = {"default": "average"}
            hip.input_pointcloud(p, ["x", "y", "z", "feat"], R, t.copy() + hip.center, 1.0, 1.0)
            orc.update_map_with_kernel(p, R, t, 1.0, 1.0)
            orc.semantic_update(p, R, t, average=[(3, 0)])
            for k in range(6):
                hip.update_time(); orc.update_time()
        assert_planes_close(hip.elevation_map, orc.elevation_map, what=scatter)
        assert np.allclose(hip.semantic_map.semantic_map[0], orc.semantic_map[0], atol=1e-6, rtol=1e-5)
        res[scatter] = hip.elevation_map.tobytes() + hip.normal_map.tobytes() + hip.semantic_map.semantic_map.tobytes()
        hip.close()
    assert res["binned"] == res["atomic"]


@pytest.mark.parametrize("rays", [True, False])
@pytest.mark.parametrize("stack", [2, 8])
def test_stacked_bins_of_large_maps_are_bit_identical(stack, rays, weights):
    """maps with more than 16384 tiles (e.g. 8192^2 on one GPU) sort into bins of 2, 4, ... stacked tiles and reduce them one
    tile per workgroup (blockIdx.y); forced here on a small map (202 rows is not a multiple of 16 * stack) incl. semantic + RGB."""
    C, N = 202, 60000
    cfg = dict(eo.YAML, enable_visibility_cleanup=rays)
    res = []
    for scatter, st in (("atomic", 0), ("binned", 0), ("binned", stack)):
        hip, _ = make_pair(cfg, C, "reference_fp16", weights)
        hip.set_scatter_mode(scatter, bin_stack=st)
        hip.param.pointcloud_channel_fusions = {"rgb": "color", "default": "average"}
        R, t = fx.POSES["rotated"]
        for f, dz in enumerate((0.0, -0.02, -0.2)):
            p = fx.semantic_cloud(C, N, f); p[:, 2] += dz
            p[300:900, :2] = p[300, :2]
            hip.input_pointcloud(p, ["x", "y", "z", "s0", "s1", "c0", "rgb"], R, t.copy() + hip.center, 1.0, 1.0)
            for k in range(6):
                hip.update_time()
        res.append(hip.elevation_map.tobytes() + hip.normal_map.tobytes() + hip.semantic_map.semantic_map.tobytes())
        hip.close()
    assert res[0] == res[1] == res[2]


@pytest.mark.parametrize("d", [1, 3, 10])
def test_fused_post_kernel_equals_the_two_stencil_stages(d, weights):
    """k_post (dilation -> traversability + normals in one launch) vs k_dilate + k_trav_normal, holes everywhere."""
    C = 202
    cfg = dict(eo.YAML, dilation_size=d)
    a, _ = make_pair(cfg, C, "reference_fp16", weights)
    b, orc = make_pair(cfg, C, "reference_fp16", weights)
    rng = np.random.default_rng(d)
    e = np.zeros((7, C, C), np.float32)
    e[5] = rng.uniform(-1, 1, (C, C)); e[2] = rng.uniform(0, 1, (C, C)) < 0.6; e[6] = rng.uniform(0, 1, (C, C)) < 0.1
    e[2][:, :5] = rng.uniform(0, 1, (C, 5)) < 0.5; e[2][40:80, 50:120] = 0; e[6][40:80, 50:120] = 0
    e[2][0, :] = 1; e[2][:, C - 1] = 1            # "valid" border cells are never a source, but keep their own value
    a.elevation_map = e; b.elevation_map = e
    orc.elevation_map[...] = e
    a.stage("dilate"); a.stage("traversability_normals")
    b.stage("post")
    orc.dilate(); orc.traversability(); orc.normals()
    assert np.array_equal(a.traversability_input, b.traversability_input)
    assert a.elevation_map.tobytes() == b.elevation_map.tobytes()
    assert a.normal_map.tobytes() == b.normal_map.tobytes()
    assert np.array_equal(b.traversability_input, orc.traversability_input)
    assert_planes_close(b.elevation_map, orc.elevation_map, what="post")


def test_fp32_index_mode_on_a_map_beyond_the_half_range(weights):
    """cell_n > 2049: the reference's half-precision clamp cannot address the map; index_mode fp32 (same source with
    float16 := float) is the defined behaviour there.  Also exercises the auto mode selection and > 1024 tiles."""
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    C, N = 2560, 300000
    cfg = dict(eo.DEFAULTS); cfg.update(eo.YAML); cfg.update(enable_visibility_cleanup=False)
    hip = ElevationMap(parameter_from(cfg, C, "auto", weights))
    assert hip.index_mode == "fp32"
    orc = eo.OracleMap(eo.make_params(cfg, cell_n=C, mode="fp32", weights=weights))
    R, t = fx.POSES["rotated"]
    for f, dz in enumerate((0.0, -0.05)):
        p = fx.cloud(C, N, f, dz=dz)
        hip.bind_points(p)
        i1 = hip.point_index(R, t); i0 = orc.point_index(p, R, t)
        assert all(np.array_equal(a, b) for a, b in zip(i1, i0))
        hip.update_map_with_kernel(None, [], R, t.copy(), 1.0, 1.0)
        orc.update_map_with_kernel(p, R, t, 1.0, 1.0)
        hip.update_time(); orc.update_time()
    assert_planes_close(hip.elevation_map, orc.elevation_map, what="fp32 2560")
    with pytest.raises(Exception):
        ElevationMap(parameter_from(cfg, C, "reference_fp16", weights))


@pytest.mark.parametrize("d", [12, 20, 32])
def test_large_dilation_radius(d, weights):
    """dilation radii whose LDS tiles exceed the default 64 KB window (emap_create accepts up to 32): fused and staged stencils"""
    C, N = 202, 6000                                     # sparse cloud: large unknown regions, long-range fills
    cfg = dict(eo.YAML, dilation_size=d, enable_visibility_cleanup=False)
    hip, orc = make_pair(cfg, C, "reference_fp16", weights)
    R, t = fx.POSES["rotated"]
    p = fx.cloud(C, N, 3)
    hip.update_map_with_kernel(p, [], R, t.copy(), 0.0, 0.0)          # k_post
    orc.update_map_with_kernel(p, R, t, 0.0, 0.0)
    assert np.array_equal(hip.traversability_input, orc.traversability_input)
    assert_planes_close(hip.elevation_map, orc.elevation_map, what="d=%d" % d)
    hip.stage("dilate")                                                 # k_dilate on the same state
    assert np.array_equal(hip.traversability_input, orc.traversability_input)
