"""GPU: the RGB / semantic fusion declared for a frame (emap_frame_semantics, round 6).  A tile-binned frame without a visibility pass
sorts 32-byte records that CARRY the cloud's channel columns and fuses them in the tile kernel that fuses the heights; every other frame
runs the stand-alone kernels before emap_update returns.  Whatever the form: the layers an emap_semantic_update call behind the frame
leaves (reference EM/elevation_mapping.py:366-368, EM/semantic_map.py:223-259) -- compared with that call BIT FOR BIT and with the
oracle."""
import ctypes as ct

import numpy as np
import pytest

import _fixtures as fx
from _util import assert_planes_equal, make_pair
from oracle import emap_oracle as eo

pytestmark = pytest.mark.gpu
CH = ["x", "y", "z", "s0", "s1", "c0", "rgb"]
FUSIONS = {"rgb": "color", "c0": "class_average", "default": "average"}
NO_RAYS = dict(eo.YAML, enable_visibility_cleanup=False)


def _pair(C, mode="reference_fp16", cfg=NO_RAYS, fusions=FUSIONS):
    hip, orc = make_pair(cfg, C, mode)
    hip.param.pointcloud_channel_fusions = dict(fusions)
    return hip, orc


def _separate_frame(hip, p, R, t, pn, on):
    """the same frame WITHOUT a declaration: emap_update, then the stand-alone fusion (what every frame did until round 5)"""
    hip.bind_points(p)
    hip.semantic_map.prepare(CH[3:])
    hip.update_map_with_kernel(None, [], R, t.copy(), pn, on)
    hip.semantic_map.update_layers_pointcloud(hip, CH[3:], R, np.asarray(t, np.float32) - hip.center)


def _sem_equal(a, b, what):
    assert_planes_equal(a.semantic_map.semantic_map, b.semantic_map.semantic_map, names=a.semantic_map.layer_names, what=what)
    assert_planes_equal(a.elevation_map, b.elevation_map, what=what)


@pytest.mark.parametrize("mode", ["reference_fp16", "fp32"])
@pytest.mark.parametrize("stack", [0, 2, 4])
@pytest.mark.parametrize("noise", [0.0, 1.0])
def test_in_tile_pass_equals_the_separate_call_and_the_oracle(mode, stack, noise):
    """stack: bins of 2 / 4 stacked tiles as maps beyond 16384 tiles sort (the siblings of a bin filter the bin's records); noise 1.0:
    the drift gate is open (k_tile_count reads the leading 16 bytes of the 32-byte records), 0.0: host-decidably shut"""
    C, N = 200, 60000
    one, orc = _pair(C, mode)
    two, _ = _pair(C, mode)
    for h in (one, two):
        h.set_scatter_mode("binned", stack)
    R, t = fx.POSES["rotated"]
    for f in range(3):
        p = fx.semantic_cloud(C, N, f)
        one.input_pointcloud(p, CH, R, t.copy(), noise, noise)
        # (noise 1.0, stack 4: a BIN of four stacked tiles holds more than SPLIT_CAP records from the second frame on -- heavy-tile parts
        # in the launch, the stand-alone kernel reads the channels from the 32-byte records)
        assert one.last_update_path() == "binned" and one.last_frame_semantics() == ("in_tile_pass" if f == 0 or noise == 0.0 or stack < 4 else "carried")
        _separate_frame(two, p, R, t, noise, noise)
        orc.update_map_with_kernel(p, R, t, noise, noise)
        orc.semantic_update(p, R, t, average=[(3, 0), (4, 1)], class_average=[(5, 2)], color=[(6, 3)], alpha=0.5)
        for h in (one, two, orc):
            h.update_time()
        _sem_equal(one, two, "frame %d" % f)
    sm = one.semantic_map.semantic_map
    assert np.allclose(sm[:3], orc.semantic_map[:3], atol=1e-6, rtol=1e-5)
    assert np.array_equal(sm[3].view(np.uint32), orc.semantic_map[3].view(np.uint32))
    assert_planes_equal(one.elevation_map, orc.elevation_map, what="heights of carrying frames")
    assert int((sm[3].view(np.uint32) != 0).sum()) > 1000 and int((sm[0] != 0).sum()) > 1000


def test_float64_host_clouds_and_interleaved_device_clouds_carry_alike():
    """the upload path de-interleaves (one aligned 16-byte channel row per point); a device cloud bound as interleaved (N, 7) rows is read
    column by column: the same records"""
    C, N = 130, 39999
    a, _ = _pair(C)
    b, _ = _pair(C)
    for h in (a, b):
        h.set_scatter_mode("binned")
    R, t = fx.POSES["identity"]
    import bench
    hip_rt = bench.Hip()
    for f in range(2):
        p = fx.semantic_cloud(C, N, f)
        a.input_pointcloud(p.astype(np.float64), CH, R, t.copy(), 1.0, 1.0)
        d = hip_rt.malloc(p.nbytes); hip_rt.h2d(d, p)
        b.bind_points_device(d.value, N, 7)
        b.update_map_with_kernel(None, CH[3:], R, t.copy(), 1.0, 1.0)
        assert a.last_frame_semantics() == b.last_frame_semantics() == "in_tile_pass"
        b.sync(); hip_rt.free(d)
        _sem_equal(a, b, "frame %d" % f)


def test_two_channel_cloud_reads_missing_columns_as_zero():
    """a cloud with fewer than four extra columns: the record's unused slots are never read past the row"""
    C, N = 130, 39999
    chs = ["x", "y", "z", "s0", "rgb"]
    fus = {"rgb": "color", "default": "average"}
    one, orc = _pair(C, fusions=fus)
    one.set_scatter_mode("binned")
    R, t = fx.POSES["identity"]
    for f in range(2):
        p7 = fx.semantic_cloud(C, N, f)
        p = np.ascontiguousarray(p7[:, [0, 1, 2, 3, 6]])
        one.input_pointcloud(p, chs, R, t.copy(), 0.0, 0.0)
        assert one.last_frame_semantics() == "in_tile_pass"
        orc.update_map_with_kernel(p, R, t, 0.0, 0.0)
        orc.semantic_update(p, R, t, average=[(3, 0)], class_average=[], color=[(4, 1)], alpha=0.5)
    sm = one.semantic_map.semantic_map
    assert np.allclose(sm[0], orc.semantic_map[0], atol=1e-6, rtol=1e-5)
    assert np.array_equal(sm[1].view(np.uint32), orc.semantic_map[1].view(np.uint32))


def test_frames_that_cannot_carry_fuse_the_same_layers():
    """a visibility pass in the frame (k_rays walks 16-byte records), the atomic path, more than four channel columns: the
    declaration falls back to the stand-alone kernels inside the frame -- same layers as the call behind the frame"""
    C, N = 130, 30000
    R, t = fx.POSES["identity"]
    for cfg, scatter in ((eo.YAML, "binned"), (NO_RAYS, "atomic")):
        one, _ = _pair(C, cfg=cfg)
        two, _ = _pair(C, cfg=cfg)
        for h in (one, two):
            h.set_scatter_mode(scatter)
        for f in range(2):
            p = fx.semantic_cloud(C, N, f)
            one.input_pointcloud(p, CH, R, t.copy(), 1.0, 1.0)
            assert one.last_frame_semantics() == "separate"
            _separate_frame(two, p, R, t, 1.0, 1.0)
            _sem_equal(one, two, "%s frame %d" % (scatter, f))
    # six averaged channels: more than a 32-byte record carries
    wide = ["x", "y", "z", "s0", "u0", "u1", "u2", "u3", "s1"]
    one, orc = _pair(C, fusions={"default": "average"})
    one.set_scatter_mode("binned")
    rng = np.random.default_rng(5)
    for f in range(2):
        p = np.concatenate([fx.cloud(C, N, f), rng.uniform(0, 1, (N, 6)).astype(np.float32)], axis=1)
        one.input_pointcloud(p, wide, R, t.copy(), 0.0, 0.0)
        assert one.last_frame_semantics() == "separate"
        orc.update_map_with_kernel(p, R, t, 0.0, 0.0)
        orc.semantic_update(p, R, t, average=[(3 + k, k) for k in range(6)], class_average=[], color=[], alpha=0.5)
    assert one.semantic_map.layer_names == wide[3:]
    assert np.allclose(one.semantic_map.semantic_map, orc.semantic_map[:6], atol=1e-6, rtol=1e-5)


def test_heavy_tiles_read_their_channels_from_the_records():
    """a launch with heavy-tile parts keeps the stand-alone semantic kernel (its parts share sums through scratch) -- on 32-byte records,
    channels read from the record; the first frame (no parts listed yet) fuses in the tile pass"""
    C, N = 300, 200000
    one, orc = _pair(C)
    two, _ = _pair(C)
    for h in (one, two):
        h.set_scatter_mode("binned")
    R, t = fx.POSES["identity"]
    seen = set()
    for f in range(4):
        p = fx.semantic_cloud(C, N, f)
        k = int(N * (0.5 + 0.1 * f))
        p[:k, :2] *= np.float32(0.04 + 0.02 * f)                 # the central patch holds most of the cloud
        one.input_pointcloud(p, CH, R, t.copy(), 1.0, 1.0)
        one.sync()
        seen.add(one.last_frame_semantics())
        _separate_frame(two, p, R, t, 1.0, 1.0)
        two.sync()
        orc.update_map_with_kernel(p, R, t, 1.0, 1.0)
        orc.semantic_update(p, R, t, average=[(3, 0), (4, 1)], class_average=[(5, 2)], color=[(6, 3)], alpha=0.5)
        for h in (one, two, orc):
            h.update_time()
        _sem_equal(one, two, "heavy frame %d" % f)
    assert seen == {"in_tile_pass", "carried"}, seen
    sm = one.semantic_map.semantic_map
    assert np.allclose(sm[:3], orc.semantic_map[:3], atol=1e-6, rtol=1e-5)
    assert np.array_equal(sm[3].view(np.uint32), orc.semantic_map[3].view(np.uint32))


def test_map_moves_between_carrying_frames():
    """pending map shifts are written out by the tile kernel's rewrite; the semantic planes were cleared band-wise by the shift"""
    C, N = 200, 60000
    one, _ = _pair(C)
    two, _ = _pair(C)
    for h in (one, two):
        h.set_scatter_mode("binned")
    R, t = fx.POSES["identity"]
    for f in range(4):
        p = fx.semantic_cloud(C, N, f)
        one.input_pointcloud(p, CH, R, t.copy(), 1.0, 1.0)
        _separate_frame(two, p, R, t, 1.0, 1.0)
        _sem_equal(one, two, "frame %d" % f)
        for h in (one, two):
            h.move_to(np.array([0.04 * 7 * (f + 1), -0.04 * 5 * (f + 1), 0.02 * f], np.float32), np.eye(3))


def test_a_declaration_lasts_one_frame_and_bad_columns_are_rejected_by_that_frame():
    C, N = 130, 30000
    hip, _ = _pair(C)
    hip.set_scatter_mode("binned")
    lib, ctx = hip._lib, hip._ctx
    R, t = fx.POSES["identity"]
    p = fx.semantic_cloud(C, N, 0)
    hip.input_pointcloud(p, CH, R, t.copy(), 0.0, 0.0)
    before = hip.semantic_map.semantic_map
    hip.update_map_with_kernel(fx.semantic_cloud(C, N, 1), [], R, t.copy(), 0.0, 0.0)          # no channels: nothing declared, nothing fused
    assert np.array_equal(before, hip.semantic_map.semantic_map)
    from elevation_mapping_cupy_amd._lib import EmapSemSpec, f32p
    spec = EmapSemSpec(); spec.n_sum = 1; spec.sum_chan[0] = 9; spec.sum_layer[0] = 0          # the cloud has 7 columns
    assert lib.emap_frame_semantics(ctx, ct.byref(spec), 0) == 0
    Rf = np.ascontiguousarray(R, np.float32).reshape(9); tf = np.ascontiguousarray(t, np.float32).reshape(3)
    assert lib.emap_update(ctx, f32p(Rf), f32p(tf), ct.c_double(0.0), ct.c_double(0.0), None) != 0
    assert b"channel" in lib.emap_last_error(ctx)
    assert lib.emap_update(ctx, f32p(Rf), f32p(tf), ct.c_double(0.0), ct.c_double(0.0), None) == 0      # the declaration died with its frame
    assert np.array_equal(before, hip.semantic_map.semantic_map)
