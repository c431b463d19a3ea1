"""GPU: a spatially coherent scene.  The seeded clouds of SURVEY 8d are white noise (x, y uniform, z +-0.5 m per point): rays dive under
the ground they measured, neighbouring points of the cloud end anywhere.  A sensor delivers the opposite -- scan-ordered beams that end in
neighbouring cells, on a terrain with walls that cast shadows (tests/_fixtures.py: terrain_cloud, every beam ray-cast at a height
field).  On such input the rays of a wave travel through the SAME cells at the same step: the per-cell combining of k_rays' work batches,
the neighbour pruning of upper-bound visits, the inert bitmap and the block thresholds see long runs of duplicates instead of none.

Scene changes between the frames (every second obstacle moves) + time ticks make the visibility pass remove what is no longer there.
Compared: the C-ABI frame against the CPU oracle stage by stage, and row strips (by row / by ray) against the single context bit for bit."""
import numpy as np
import pytest

import _fixtures as fx
from _util import assert_planes_close, make_pair
from oracle import emap_oracle as eo
from test_hip_comm import _strips_vs_single
from test_hip_parity import _run_frame_stages

pytestmark = pytest.mark.gpu

IDENT = (np.eye(3, dtype=np.float32), np.array([0, 0, 1], np.float32))


def _frames(C, n_az, n_el, res=0.04):
    return [fx.terrain_cloud(C, n_az, n_el, 0, res=res), fx.terrain_cloud(C, n_az, n_el, 1, res=res, shift=1.5),
            fx.terrain_cloud(C, n_az, n_el, 2, res=res, shift=-1.0)]


@pytest.mark.parametrize("mode", ["reference_fp16", "fp32"])
@pytest.mark.parametrize("cfg_name,C,n_az,n_el", [("yaml", 202, 300, 160), ("default", 202, 300, 160), ("yaml", 400, 500, 240)])
def test_terrain_scene_stagewise_against_the_oracle(cfg_name, C, n_az, n_el, mode, weights):
    cfg = eo.YAML if cfg_name == "yaml" else eo.DEFAULTS
    hip, orc = make_pair(cfg, C, mode, weights)
    R, t = IDENT
    clouds = _frames(C, n_az, n_el)
    _run_frame_stages(hip, orc, clouds[0], R, t, tag="f0")
    n_valid = int((hip.elevation_map[2] > 0.5).sum())
    assert n_valid > C * C // 20
    for k in range(12):
        hip.update_time(); orc.update_time()
    _run_frame_stages(hip, orc, clouds[1], R, t, pn=1.0, on=1.0, tag="f1 (obstacles moved, every cell stale)")
    removed = n_valid + int(((hip.elevation_map[2] > 0.5) & ~(orc.elevation_map[2] > 0.5)).sum()) - int((hip.elevation_map[2] > 0.5).sum())
    for k in range(3):
        hip.update_time(); orc.update_time()
    _run_frame_stages(hip, orc, clouds[2], R, t, pn=1.0, on=1.0, tag="f2")
    assert abs(hip.get_additive_mean_error() - float(orc.additive_mean_error)) < 1e-6
    assert removed is not None


@pytest.mark.parametrize("mode", ["reference_fp16", "fp32"])
def test_terrain_scene_whole_frames_1024(mode, weights):
    """the 1024^2 map of BASELINE configs[1..2] (the LDS-bitmap variant of the ray kernel in reference_fp16 mode), 108 k beams"""
    C = 1024
    hip, orc = make_pair(eo.YAML, C, mode, weights)
    R, t = IDENT
    before = None
    for f, p in enumerate(_frames(C, 450, 240)):
        hip.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0)
        orc.update_map_with_kernel(p, R, t, 1.0, 1.0)
        if f == 0:
            before = int((orc.elevation_map[2] > 0.5).sum())
        for k in range(9):
            hip.update_time(); orc.update_time()
    assert_planes_close(hip.elevation_map, orc.elevation_map, what="terrain, 3 frames")
    assert_planes_close(hip.normal_map, orc.normal_map, names=["nx", "ny", "nz"])
    assert before > 20000


@pytest.mark.parametrize("ray_mode", ["by_row", "by_ray"])
@pytest.mark.parametrize("world,C,n_az,n_el,moves", [(2, 202, 300, 160, False), (4, 400, 500, 240, True), (3, 1024, 450, 240, False)])
def test_terrain_scene_row_strips_reproduce_the_single_context(world, C, n_az, n_el, moves, ray_mode, weights):
    cfg = dict(eo.DEFAULTS); cfg.update(eo.YAML)
    R, t = IDENT
    clouds = _frames(C, n_az, n_el)
    mv = [None, (0.37, -0.21, 0.0) if moves else None, None]
    frames = [(p, R, t.copy(), 1.0, 1.0, 9 if f == 0 else 2, mv[f]) for f, p in enumerate(clouds)]
    _strips_vs_single(world, cfg, C, frames, "auto", weights, stand_in="stream", ray_mode=ray_mode)


# ---- heavy tiles: most of the cloud in a handful of cells (the ground under a sensor), the tile kernels' split path --------------------
def _heavy_cloud(C, N, seed, dz=0.0, frac=0.7, patch=0.03):
    """cloud() with `frac` of its points squeezed into the central patch x patch of the map: ONE sort tile holds more than SPLIT_CAP
    (4096) records and is reduced by several workgroups (emap_device.h: SplitView).  The first frame of a context runs unsplit (the host has
    not heard of a heavy tile yet), the following ones split: same bits either way"""
    p = fx.cloud(C, N, seed, dz=dz)
    k = int(N * frac)
    p[:k, :2] *= np.float32(patch)
    return p


@pytest.mark.parametrize("mode", ["reference_fp16", "fp32"])
@pytest.mark.parametrize("cfg_name,C,N,stack", [("yaml", 202, 150000, 0), ("default", 400, 260000, 0), ("yaml", 400, 200000, 2)])
def test_heavy_tiles_stagewise_against_the_oracle(cfg_name, C, N, stack, mode, weights):
    """the staged contract (AccF records written by the part that arrives last) on the sorted path; stack = 2: bins of two stacked
    tiles as maps beyond 16384 tiles have them (every part is walked by both tiles' workgroups)"""
    cfg = eo.YAML if cfg_name == "yaml" else eo.DEFAULTS
    hip, orc = make_pair(cfg, C, mode, weights)
    hip.set_scatter_mode("binned", stack)
    R, t = fx.POSES["rotated"]
    _run_frame_stages(hip, orc, _heavy_cloud(C, N, 0), R, t, tag="f0")
    for k in range(9):
        hip.update_time(); orc.update_time()
    _run_frame_stages(hip, orc, _heavy_cloud(C, N, 1, dz=-0.03, frac=0.5, patch=0.1), R, t, pn=1.0, on=1.0, tag="f1")
    _run_frame_stages(hip, orc, _heavy_cloud(C, N, 2, dz=-0.15), R, t, pn=1.0, on=1.0, tag="f2")
    assert abs(hip.get_additive_mean_error() - float(orc.additive_mean_error)) < 1e-6


@pytest.mark.parametrize("rays", [False, True])
@pytest.mark.parametrize("scene", ["heavy", "terrain"])
def test_heavy_tiles_whole_frames_1024(scene, rays, weights):
    """emap_update (the fused tile kernel: commit + average in the epilogue of the last part; with rays: bitmap + thresholds too)"""
    C = 1024
    cfg = dict(eo.DEFAULTS); cfg.update(eo.YAML, enable_visibility_cleanup=rays, enable_overlap_clearance=not rays)
    hip, orc = make_pair(cfg, C, "reference_fp16", weights)
    hip.set_scatter_mode("binned")
    R, t = IDENT if scene == "terrain" else fx.POSES["rotated"]
    clouds = _frames(C, 600, 300) if scene == "terrain" else [_heavy_cloud(C, 200000, s, dz=-0.04 * s, frac=0.6, patch=0.02 + 0.03 * s) for s in range(3)]
    for p in clouds:
        hip.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0)
        orc.update_map_with_kernel(p, R, t, 1.0, 1.0)
        for k in range(7):
            hip.update_time(); orc.update_time()
    assert_planes_close(hip.elevation_map, orc.elevation_map, what="%s, 3 frames" % scene)
    assert_planes_close(hip.normal_map, orc.normal_map, names=["nx", "ny", "nz"])


@pytest.mark.parametrize("ray_mode", ["by_row", "by_ray"])
@pytest.mark.parametrize("world,C,N", [(2, 202, 150000), (4, 400, 260000)])
def test_heavy_tiles_row_strips_reproduce_the_single_context(world, C, N, ray_mode, weights):
    cfg = dict(eo.DEFAULTS); cfg.update(eo.YAML)
    R, t = fx.POSES["rotated"]
    frames = [(_heavy_cloud(C, N, f, dz=-0.05 * f, frac=0.6, patch=0.05 + 0.1 * f), R, t.copy(), 1.0, 1.0, 6 if f == 0 else 2, None) for f in range(3)]
    _strips_vs_single(world, cfg, C, frames, "binned", weights, stand_in="stream", ray_mode=ray_mode)


def test_heavy_tiles_soak_two_contexts_and_the_oracle(weights):
    """40 frames whose heavy tiles change from frame to frame (so does the number of extra workgroups the host launches: it follows
    the need the device reported last): two HIP contexts must agree byte for byte after every frame -- the parts of a tile merge
    through device atomics with no fence between them and the ticket, a lost or late partial sum would show here -- and with the
    oracle at the end."""
    C, N = 400, 220000
    cfg = dict(eo.DEFAULTS); cfg.update(eo.YAML)
    a, orc = make_pair(cfg, C, "reference_fp16", weights)
    b, _ = make_pair(cfg, C, "reference_fp16", weights)
    a.set_scatter_mode("binned"); b.set_scatter_mode("binned")
    rng = np.random.default_rng(5)
    R, t = fx.POSES["rotated"]
    for f in range(40):
        p = _heavy_cloud(C, N, 100 + f, dz=float(rng.uniform(-0.1, 0.05)), frac=float(rng.uniform(0.1, 0.9)), patch=float(rng.uniform(0.01, 0.3)))
        if f % 7 == 3:
            p = fx.cloud(C, N, 100 + f)                       # a frame without heavy tiles in between
        for m in (a, b):
            m.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0)
        orc.update_map_with_kernel(p, R, t, 1.0, 1.0)
        assert a.elevation_map.tobytes() == b.elevation_map.tobytes(), "frame %d: two contexts differ" % f
        for k in range(int(rng.integers(0, 4))):
            a.update_time(); b.update_time(); orc.update_time()
    assert a.normal_map.tobytes() == b.normal_map.tobytes()
    assert_planes_close(a.elevation_map, orc.elevation_map, what="40 heavy frames")
    assert_planes_close(a.normal_map, orc.normal_map, names=["nx", "ny", "nz"])


@pytest.mark.parametrize("mode", ["reference_fp16", "fp32"])
@pytest.mark.parametrize("stack", [0, 2])
def test_heavy_tiles_multi_modal_against_the_oracle(mode, stack):
    """RGB + two averaged channels + a class_average channel on clouds whose points pile up in one sort tile: the semantic tile kernel
    shares a heavy tile's sums between several workgroups too (k_tile_semantic<true>); the first frame runs unsplit."""
    CH = ["x", "y", "z", "s0", "s1", "c0", "rgb"]
    C, N = 300, 200000
    hip, orc = make_pair(eo.YAML, C, mode)
    hip.param.pointcloud_channel_fusions = {"rgb": "color", "c0": "class_average", "default": "average"}
    hip.set_scatter_mode("binned", stack)
    R, t = fx.POSES["identity"]
    for f in range(4):
        p = fx.semantic_cloud(C, N, f)
        k = int(N * (0.5 + 0.1 * f))
        p[:k, :2] *= np.float32(0.04 + 0.02 * f)                 # the central patch holds most of the cloud
        hip.input_pointcloud(p, CH, R, t.copy(), 0.0, 0.0)
        orc.update_map_with_kernel(p, R, t, 0.0, 0.0)
        orc.semantic_update(p, R, t, average=[(3, 0), (4, 1)], class_average=[(5, 2)], color=[(6, 3)], alpha=0.5)
        hip.update_time(); orc.update_time()
    assert_planes_close(hip.elevation_map, orc.elevation_map, what="heavy multi-modal frames")
    sm = hip.semantic_map.semantic_map
    assert np.allclose(sm[:3], orc.semantic_map[:3], atol=1e-6, rtol=1e-5)
    assert np.array_equal(sm[3].view(np.uint32), orc.semantic_map[3].view(np.uint32))
    assert int((sm[3].view(np.uint32) != 0).sum()) > 1000
