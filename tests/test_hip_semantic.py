"""GPU: RGB / semantic point-cloud fusion through the reference's API surface (input_pointcloud with extra channels,
pointcloud_channel_fusions mapping) vs the oracle and vs the reference kernels' golden output."""
import os

import numpy as np
import pytest

import _fixtures as fx
from _util import make_pair
from oracle import emap_oracle as eo

pytestmark = pytest.mark.gpu
CH = ["x", "y", "z", "s0", "s1", "c0", "rgb"]
FUSIONS = {"rgb": "color", "c0": "class_average", "default": "average"}


def _hip(C, mode="reference_fp16"):
    hip, orc = make_pair(eo.YAML, C, mode)
    hip.param.pointcloud_channel_fusions = dict(FUSIONS)
    return hip, orc


@pytest.mark.parametrize("scatter", ["atomic", "binned"])
def test_against_reference_golden(scatter):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "semantic_yaml66.npz"))
    C, N = 66, 6000
    hip, _ = _hip(C)
    hip.set_scatter_mode(scatter)
    R, t = fx.POSES["rotated"]
    p = fx.semantic_cloud(C, N, 5)
    hip.semantic_map.prepare(CH[3:])
    assert hip.semantic_map.layer_names == ["s0", "s1", "c0", "rgb"]
    hip.semantic_map.set_layer("c0", fx.semantic_prev(C))
    hip.input_pointcloud(p.astype(np.float64), CH, R, t.copy(), 0.0, 0.0)       # the ROS wrapper hands over float64
    sm = hip.semantic_map.semantic_map
    assert np.allclose(sm[:3], g["sem"][:3], atol=1e-6, rtol=1e-6)
    assert np.array_equal(sm[3].view(np.uint32), g["sem"][3].view(np.uint32))
    assert hip.exists_layer("rgb") and hip.exists_layer("elevation") and not hip.exists_layer("nope")
    out = np.zeros((C - 2, C - 2), np.float32)
    hip.get_map_with_name_ref("s0", out)
    assert np.array_equal(out, np.flip(sm[0][1:-1, 1:-1]))


@pytest.mark.parametrize("scatter", ["atomic", "binned"])
@pytest.mark.parametrize("mode", ["reference_fp16", "fp32"])
def test_three_frames_against_oracle(mode, scatter):
    C, N = 130, 30000
    hip, orc = _hip(C, mode)
    hip.set_scatter_mode(scatter)
    R, t = fx.POSES["identity"]
    for f in range(3):
        p = fx.semantic_cloud(C, N, f)
        hip.input_pointcloud(p, CH, R, t.copy(), 0.0, 0.0)
        orc.update_map_with_kernel(p, R, t, 0.0, 0.0)
        orc.semantic_update(p, R, t, average=[(3, 0), (4, 1)], class_average=[(5, 2)], color=[(6, 3)], alpha=0.5)
        hip.update_time(); orc.update_time()
    sm = hip.semantic_map.semantic_map
    assert np.allclose(sm[:3], orc.semantic_map[:3], atol=1e-6, rtol=1e-5)
    assert np.array_equal(sm[3].view(np.uint32), orc.semantic_map[3].view(np.uint32))


def test_shift_moves_semantic_layers_with_the_map():
    C, N = 66, 6000
    hip, _ = _hip(C)
    R, t = fx.POSES["identity"]
    hip.input_pointcloud(fx.semantic_cloud(C, N, 1), CH, R, t.copy(), 0.0, 0.0)
    before_s, before_e = hip.semantic_map.semantic_map, hip.elevation_map
    hip.move_to(np.array([3 * 0.04, -2 * 0.04, 0.25], np.float32), np.eye(3))
    after_s, after_e = hip.semantic_map.semantic_map, hip.elevation_map
    # reference move_to: shift_map_xy(-delta_pixel) = roll by (-3, +2) with zero padding; planes 0 and 5 -= dz
    want_s = np.roll(before_s, (-3, 2), axis=(1, 2)); want_s[:, -3:, :] = 0; want_s[:, :, :2] = 0
    assert np.array_equal(after_s, want_s)
    want_e = np.roll(before_e, (-3, 2), axis=(1, 2)); want_e[:, -3:, :] = 0; want_e[:, :, :2] = 0
    want_e[1, -3:, :] = hip.initial_variance; want_e[1, :, :2] = hip.initial_variance
    want_e[0] -= np.float32(0.25); want_e[5] -= np.float32(0.25)
    assert np.allclose(after_e, want_e, atol=1e-6)
    assert np.allclose(hip.center, [0.12, -0.08, 0.25], atol=1e-6)


def test_two_colour_channels_quirk_same_on_both_paths():
    """K = 2 colour channels: the reference launches add_color_kernel with size = N while decoding id = i / K
    (SURVEY appendix B.12); both device paths must reproduce that identically, plus 6 averaged channels (> one LDS group)."""
    C, N = 130, 20000
    res = []
    for scatter in ("atomic", "binned"):
        hip, _ = _hip(C)
        hip.set_scatter_mode(scatter)
        hip.param.pointcloud_channel_fusions = {"rgb.*": "color", "default": "average"}
        R, t = fx.POSES["rotated"]
        p = fx.cloud(C, N, 3, extra=8)
        rng = np.random.default_rng(9)
        p[:, 3] = rng.integers(0, 1 << 24, N, dtype=np.uint32).view(np.float32)
        p[:, 4] = rng.integers(0, 1 << 24, N, dtype=np.uint32).view(np.float32)
        p[: N // 4, :2] = p[0, :2]
        hip.input_pointcloud(p, ["x", "y", "z", "rgb", "rgb2", "a", "b", "c", "d", "e", "f"], R, t.copy(), 0.0, 0.0)
        sm = hip.semantic_map.semantic_map
        assert hip.semantic_map.layer_names == ["rgb", "rgb2", "a", "b", "c", "d", "e", "f"]
        res.append(sm)
    assert np.array_equal(res[0][:2].view(np.uint32), res[1][:2].view(np.uint32))
    assert np.allclose(res[0][2:], res[1][2:], atol=1e-6, rtol=1e-6)
    assert (res[0][0].view(np.uint32) != 0).sum() > 100


BAYES_CH = ["x", "y", "z", "p0", "p1", "b0", "rgb"]
BAYES_FUSIONS = {"p[01]": "class_bayesian", "b0": "bayesian_inference", "rgb": "color"}


@pytest.mark.parametrize("scatter", ["atomic", "binned"])
def test_bayesian_fusions_against_reference_golden(scatter):
    """class_bayesian (K = 2: launch-size quirk, theta < 0 ignored, persistent pseudo-counts, renormalisation) and
    bayesian_inference against the output of the reference's own kernels (tests/golden/make_golden.py: bayes66)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "bayes_yaml66.npz"))
    C, N = 66, 6000
    hip, _ = make_pair(eo.YAML, C, "reference_fp16")
    hip.param.pointcloud_channel_fusions = dict(BAYES_FUSIONS)
    hip.set_scatter_mode(scatter)
    R, t = fx.POSES["rotated"]
    p = fx.bayes_cloud(C, N, 5)
    hip.semantic_map.prepare(BAYES_CH[3:])
    assert hip.semantic_map.layer_names == ["p0", "p1", "b0", "rgb"]
    prior = fx.bayes_alpha_prior(C)
    hip.semantic_map.set_alpha("p0", prior[0]); hip.semantic_map.set_alpha("p1", prior[1])
    hip.semantic_map.set_layer("b0", fx.semantic_prev(C))
    hip.input_pointcloud(p, BAYES_CH, R, t.copy(), 0.0, 0.0)
    sm = hip.semantic_map.semantic_map
    alpha = np.stack([hip.semantic_map.get_alpha("p0"), hip.semantic_map.get_alpha("p1")])
    assert np.allclose(alpha, g["alpha"], atol=1e-5, rtol=1e-5)
    assert np.allclose(sm[:2], g["sem"][:2], atol=1e-6, rtol=1e-5)
    assert np.array_equal(sm[2], g["sem"][2])
    assert (sm[3].view(np.uint32) != 0).sum() > 100          # the colour channel of the same cloud is fused alongside


@pytest.mark.parametrize("scatter", ["atomic", "binned"])
def test_class_bayesian_accumulates_over_frames_and_moves_with_the_map(scatter):
    C, N = 130, 30000
    hip, orc = make_pair(eo.YAML, C, "fp32")
    hip.param.pointcloud_channel_fusions = dict(BAYES_FUSIONS)
    hip.set_scatter_mode(scatter)
    R, t = fx.POSES["identity"]
    for f in range(3):
        p = fx.bayes_cloud(C, N, f)
        hip.input_pointcloud(p, BAYES_CH, R, t.copy(), 0.0, 0.0)
        orc.update_map_with_kernel(p, R, t, 0.0, 0.0)
        orc.semantic_update(p, R, t, class_bayesian=[(3, 0), (4, 1)], bayesian_inference=[(5, 2)], color=[(6, 3)])
        hip.update_time(); orc.update_time()
    sm = hip.semantic_map.semantic_map
    a0 = hip.semantic_map.get_alpha("p0")
    assert np.allclose(a0, orc.semantic_alpha[0], atol=1e-5, rtol=1e-5) and a0.max() > 1.0
    assert np.allclose(sm[:2], orc.semantic_map[:2], atol=1e-6, rtol=1e-5)
    tot = sm[0] + sm[1]
    assert np.allclose(tot[tot > 0], 1.0, atol=1e-6)          # a categorical distribution wherever anything was observed
    assert np.array_equal(sm[2], np.zeros_like(sm[2]))        # bayesian_inference never moves off its initial value (reference behaviour)
    # the pseudo-counts shift with the map (reference semantic_map.py:135-136) and survive clear() (:47-49 only zeroes the layers)
    hip.move_to(np.array([2 * 0.04, -3 * 0.04, 0.0], np.float32), np.eye(3))
    want = np.roll(a0, (-2, 3), axis=(0, 1)); want[-2:, :] = 0; want[:, :3] = 0
    assert np.array_equal(hip.semantic_map.get_alpha("p0"), want)
    hip.clear()
    assert np.array_equal(hip.semantic_map.get_alpha("p0"), want) and not hip.semantic_map.semantic_map.any()


@pytest.mark.parametrize("scatter", ["atomic", "binned"])
def test_device_cloud_layouts_agree(scatter):
    """the same multi-modal cloud bound three ways -- uploaded from the host (de-interleaved by emap_upload_points), device resident as
    interleaved (N, 7) rows (emap_set_points_device), device resident de-interleaved (emap_set_points_device_split) -- must give the
    same map and the same semantic layers bit for bit; a channel index beyond the cloud's columns is refused for every layout"""
    import ctypes as ct
    from elevation_mapping_cupy_amd._lib import EmapError
    C, N = 130, 30000
    R, t = fx.POSES["rotated"]
    hipl = ct.CDLL("libamdhip64.so")

    def dev(a):
        a = np.ascontiguousarray(a, np.float32)
        d = ct.c_void_p()
        assert hipl.hipMalloc(ct.byref(d), ct.c_size_t(a.nbytes)) == 0 and hipl.hipMemcpy(d, ct.c_void_p(a.ctypes.data), ct.c_size_t(a.nbytes), 1) == 0
        return d
    out = []
    for layout in ("upload", "rows", "split"):
        hip, _ = _hip(C)
        hip.set_scatter_mode(scatter)
        keep = []
        for f in range(2):
            p = fx.semantic_cloud(C, N, 20 + f)
            if layout == "upload":
                hip.bind_points(p)
            elif layout == "rows":
                d = dev(p); keep.append(d)
                hip.bind_points_device(d.value, N, 7)
            else:
                dx, dc = dev(p[:, :3]), dev(p[:, 3:]); keep += [dx, dc]
                hip.bind_points_device_split(dx.value, dc.value, N, 4)
            hip.update_map_with_kernel(None, CH[3:], R, t.copy(), 1.0, 1.0)
            hip.update_time()
        out.append((hip.elevation_map, hip.semantic_map.semantic_map))
        with pytest.raises(EmapError, match="channel/layer index"):
            from elevation_mapping_cupy_amd._lib import EmapSemSpec, f32p
            spec = EmapSemSpec(); spec.n_sum = 1; spec.sum_chan[0] = 7; spec.sum_layer[0] = 0
            Rf = np.ascontiguousarray(R, np.float32).ravel().copy()
            hip._chk(hip._lib.emap_semantic_update(hip._ctx, f32p(Rf), f32p(t.copy()), ct.byref(spec)))
        hip.close()
        for d in keep:
            hipl.hipFree(d)
    for e, s in out[1:]:
        assert e.tobytes() == out[0][0].tobytes() and s.tobytes() == out[0][1].tobytes()
    assert (out[0][1][3].view(np.uint32) != 0).sum() > 1000
