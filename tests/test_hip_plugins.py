"""GPU: plugin surface (YAML loading, arity dispatch, read-back post-processing) and the MinFilter sweeps vs the oracle."""
import numpy as np
import pytest

import _fixtures as fx
from _util import make_pair
from oracle import emap_oracle as eo

pytestmark = pytest.mark.gpu


def _holey_map(C, seed, hole_frac):
    rng = np.random.default_rng(seed)
    e = np.zeros((7, C, C), np.float32)
    e[0] = rng.uniform(-1, 1, (C, C)); e[2] = (rng.uniform(0, 1, (C, C)) > hole_frac)
    e[2][20:50, 30:70] = 0                       # a big hole: needs several sweeps
    return e


@pytest.mark.parametrize("d,iters", [(1, 30), (2, 3), (5, 5)])
def test_min_filter_matches_oracle(d, iters):
    from elevation_mapping_cupy_amd.plugins.min_filter import MinFilter
    C = 130
    hip, _ = make_pair(eo.DEFAULTS, C)
    e = _holey_map(C, d, 0.3)
    mf = MinFilter(cell_n=C, dilation_size=d, iteration_n=iters, emap=hip)
    got = mf(e, hip.layer_names, None, [])
    want, sweeps = eo.min_filter(C, d, iters, e[0], e[2])
    assert mf.sweeps_run == sweeps
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)])


@pytest.mark.parametrize("d,iters", [(1, 30), (2, 3), (3, 1)])
def test_max_filter_matches_oracle(d, iters):
    from elevation_mapping_cupy_amd.plugins.max_filter import MaxFilter
    C = 130
    hip, _ = make_pair(eo.DEFAULTS, C)
    e = _holey_map(C, d, 0.3)
    mf = MaxFilter(cell_n=C, dilation_size=d, iteration_n=iters, emap=hip)
    got = mf(e, hip.layer_names, None, [])
    want, sweeps = eo.max_filter(C, d, iters, e[0], e[2])
    assert mf.sweeps_run == sweeps
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)])


def test_smooth_filter_matches_scipy_uniform_filter():
    """SmoothFilter = uniform_filter(size=3) twice ('reflect' borders); scipy.ndimage is the published algorithm behind the
    reference's cupyx call.  Tolerance 1e-6 relative (float32 output of a double-accumulated 3-tap mean)."""
    from elevation_mapping_cupy_amd.plugins.smooth_filter import SmoothFilter
    C = 130
    hip, _ = make_pair(eo.DEFAULTS, C)
    e = _holey_map(C, 1, 0.3)
    e[0] += np.random.default_rng(5).normal(0, 1, (C, C)).astype(np.float32)
    sf = SmoothFilter(cell_n=C, input_layer_name="elevation", emap=hip)
    got = sf(e, hip.layer_names, None, [])
    want = eo.smooth_filter(e[0])
    assert np.allclose(got, want, rtol=1e-6, atol=1e-6)
    # by plugin-layer name, and the fallback to the elevation layer
    sf2 = SmoothFilter(cell_n=C, input_layer_name="min_filter", emap=hip)
    plug = np.stack([e[1]])
    assert np.allclose(sf2(e, hip.layer_names, plug, ["min_filter"]), eo.smooth_filter(e[1]), rtol=1e-6, atol=1e-4)
    sf3 = SmoothFilter(cell_n=C, input_layer_name="nope", emap=hip)
    assert np.allclose(sf3(e, hip.layer_names, plug, ["min_filter"]), want, rtol=1e-6, atol=1e-6)


def test_plugin_manager_yaml_and_readback(tmp_path):
    C = 66
    cfg = tmp_path / "plugins.yaml"
    cfg.write_text("""
min_filter:
  enable: True
  fill_nan: False
  is_height_layer: True
  layer_name: "min_filter"
  extra_params:
    dilation_size: 1
    iteration_n: 30
disabled_one:
  enable: False
  fill_nan: False
  is_height_layer: False
  layer_name: "nope"
  extra_params: {}
""")
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    p = parameter_from(eo.DEFAULTS, C)
    p.plugin_config_file = str(cfg)
    hip = ElevationMap(p)
    assert hip.plugin_manager.layer_names == ["min_filter"] and hip.exists_layer("min_filter")
    R, t = fx.POSES["identity"]
    hip.input_pointcloud(fx.cloud(C, 3000, 0), ["x", "y", "z"], R, t.copy(), 0.0, 0.0)
    hip.move_to(np.array([0.0, 0.0, 0.5], np.float32), np.eye(3))
    out = np.zeros((C - 2, C - 2), np.float32)
    hip.get_map_with_name_ref("min_filter", out)
    e = hip.elevation_map
    want, _ = eo.min_filter(C, 1, 30, e[0], e[2])
    want = np.flip(want[1:-1, 1:-1] + hip.center[2])           # is_height_layer: + center_z; both axes flipped
    assert np.allclose(out, want, equal_nan=True, atol=1e-6)
    elev = np.zeros((C - 2, C - 2), np.float32)
    hip.get_map_with_name_ref("elevation", elev)
    assert np.isnan(elev).sum() == int((e[2][1:-1, 1:-1] <= 0.5).sum())


@pytest.mark.parametrize("k,iters,reverse", [(3, 1, True), (3, 4, False), (5, 2, True), (4, 1, False)])
def test_erosion_matches_window_minimum(k, iters, reverse):
    """Erosion = 8-bit quantisation + cv2.erode(ones((k,k)), iterations) + de-quantisation; scipy.ndimage.minimum_filter with a
    constant border above the value range is the same published operation (OpenCV itself is absent: parity unpinned)."""
    from scipy import ndimage
    from elevation_mapping_cupy_amd.plugins.erosion import Erosion
    C = 98
    hip, _ = make_pair(eo.DEFAULTS, C)
    e = _holey_map(C, 1, 0.3)
    e[3] = np.random.default_rng(k).uniform(0, 1, (C, C)).astype(np.float32)
    er = Erosion(input_layer_name="traversability", kernel_size=k, iterations=iters, reverse=reverse, emap=hip)
    got = er(e, hip.layer_names, None, [], np.zeros((0, C, C), np.float32), [])
    layer = (1 - e[3]) if reverse else e[3]
    lo, hi = float(layer.min()), float(layer.max())
    q = ((layer - lo) * 255 / (hi - lo)).astype("uint8")
    for _ in range(iters):
        q = ndimage.minimum_filter(q, size=k, mode="constant", cval=255)    # window offsets -k//2 .. k-k//2-1 = cv2 anchor (k//2, k//2)
    want = q.astype(np.float32) * (hi - lo) / 255 + lo
    want = (1 - want) if reverse else want
    assert np.allclose(got, want, atol=1e-6)


def test_shipped_plugin_configuration_loads_and_publishes(tmp_path):
    """the four plugins the reference enables by default (config/core/plugin_config.yaml: min_filter -> smooth, inpainting,
    erosion of the traversability) plus an unknown one, which is reported and skipped instead of taking the others down"""
    C = 66
    cfg = tmp_path / "plugins.yaml"
    cfg.write_text("""
min_filter: {enable: True, fill_nan: False, is_height_layer: True, layer_name: "min_filter", extra_params: {dilation_size: 1, iteration_n: 30}}
smooth_filter: {enable: True, fill_nan: False, is_height_layer: True, layer_name: "smooth", extra_params: {input_layer_name: "min_filter"}}
inpainting: {enable: True, fill_nan: False, is_height_layer: True, layer_name: "inpaint", extra_params: {method: "telea"}}
erosion: {enable: True, fill_nan: False, is_height_layer: False, layer_name: "erosion", extra_params: {input_layer_name: "traversability", dilation_size: 3, iteration_n: 20, reverse: True}}
no_such_plugin: {enable: True, fill_nan: False, is_height_layer: False, layer_name: "ghost", extra_params: {}}
""")
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    p = parameter_from(eo.DEFAULTS, C)
    p.plugin_config_file = str(cfg)
    hip = ElevationMap(p)
    assert hip.plugin_manager.layer_names == ["min_filter", "smooth", "inpaint", "erosion"] and not hip.exists_layer("ghost")
    R, t = fx.POSES["identity"]
    hip.input_pointcloud(fx.cloud(C, 3000, 0), ["x", "y", "z"], R, t.copy(), 0.0, 0.0)
    out = {}
    for name in ("min_filter", "smooth", "inpaint", "erosion"):
        out[name] = np.zeros((C - 2, C - 2), np.float32)
        hip.get_map_with_name_ref(name, out[name])
    e = hip.elevation_map
    mf, _ = eo.min_filter(C, 1, 30, e[0], e[2])
    assert np.allclose(out["smooth"], np.flip(eo.smooth_filter(mf)[1:-1, 1:-1]), atol=1e-5, equal_nan=True)   # smooth reads the min_filter layer
    assert np.isfinite(out["inpaint"]).all()
    assert out["erosion"].min() >= -1e-6 and out["erosion"].max() <= 1 + 1e-6
    assert (out["erosion"] >= np.flip(e[3][1:-1, 1:-1]) - 1e-2).all()       # reverse erosion can only raise traversability (8-bit step)


@pytest.mark.parametrize("only_above", [True, False])
def test_device_side_layer_readback_matches_reference_postprocessing(only_above):
    """emap_publish_layer vs a NumPy restatement of get_map_with_name_ref (elevation_mapping.py:579-775)."""
    C = 98
    hip, _ = make_pair(dict(eo.YAML, enable_visibility_cleanup=True), C)
    hip.param.use_only_above_for_upper_bound = only_above
    R, t = fx.POSES["rotated"]
    for f in range(2):
        hip.input_pointcloud(fx.cloud(C, 4000, f, dz=-0.1 * f), ["x", "y", "z"], R, t.copy(), 0.0, 0.0)
        for _ in range(6):
            hip.update_time()
    hip.move_to(np.array([0.0, 0.0, 0.3], np.float32), np.eye(3))
    e, n, cz = hip.elevation_map, hip.normal_map, hip.center[2]
    if only_above:
        ub_ok = ((e[5] > 0.0) & (e[6] > 0.5)) | (e[2] > 0.5)
    else:
        ub_ok = (e[2] > 0.5) | (e[6] > 0.5)
    trav = np.full((C, C), np.nan, np.float32)
    trav[3:-3, 3:-3] = np.where((e[2] + e[6]) > 0.5, e[3], np.nan)[3:-3, 3:-3]
    want = {
        "elevation": np.where(e[2] > 0.5, e[0], np.nan) + cz, "variance": e[1], "time": e[4], "traversability": trav,
        "upper_bound": np.where(ub_ok, e[5], np.nan) + cz, "is_upper_bound": np.where(ub_ok, e[6], np.nan),
        "normal_x": n[0], "normal_y": n[1], "normal_z": n[2],
    }
    for name, w in want.items():
        out = np.zeros((C - 2, C - 2), np.float32)
        hip.get_map_with_name_ref(name, out)
        assert np.allclose(out, np.flip(w[1:-1, 1:-1]).astype(np.float32), equal_nan=True, atol=1e-6), name
        out64 = np.zeros((C - 2, C - 2), np.float64)       # non-float32 buffers take the host path
        hip.get_map_with_name_ref(name, out64)
        assert np.allclose(out64, out, equal_nan=True), name


@pytest.mark.parametrize("method", ["telea", "front"])
def test_inpainting_properties(method):
    """OpenCV's Telea arithmetic is unpinned (absent third-party library); both fills -- the fast-marching restatement on the host
    (tests/test_inpaint_telea.py) and the device-side front propagation -- must satisfy what any inpainting of the reference's 8-bit
    pipeline satisfies."""
    from elevation_mapping_cupy_amd.plugins.inpainting import Inpainting
    C = 130
    hip, _ = make_pair(eo.DEFAULTS, C)
    rng = np.random.default_rng(0)
    xx, yy = np.meshgrid(np.arange(C), np.arange(C), indexing="ij")
    e = np.zeros((7, C, C), np.float32)
    e[0] = (0.5 * np.sin(xx / 17.0) + 0.3 * np.cos(yy / 11.0)).astype(np.float32)
    e[2] = rng.uniform(0, 1, (C, C)) > 0.3
    e[2][40:70, 50:90] = 0
    e[0][e[2] < 0.5] = 0.0                                  # unknown cells hold garbage
    ip = Inpainting(cell_n=C, emap=hip, method=method)
    out = np.asarray(ip(e, hip.layer_names, None, []), np.float64)
    known = e[2] >= 0.5
    hmin, hmax = e[0][known].min(), e[0][known].max()
    step = (hmax - hmin) / 255
    assert np.isfinite(out).all()
    assert np.abs(out[known] - e[0][known]).max() <= step + 1e-6            # 8-bit round trip on known cells
    assert out.min() >= hmin - 1e-6 and out.max() <= hmax + 1e-6             # stays inside the range of the known data
    truth = (0.5 * np.sin(xx / 17.0) + 0.3 * np.cos(yy / 11.0))
    assert np.abs(out[~known] - truth[~known]).mean() < 0.05                 # and is a sensible reconstruction of the smooth surface
    assert method == "telea" or 1 <= ip.sweeps_run <= 40


def test_builtin_plugins_read_the_live_map_on_the_device(weights, tmp_path):
    """get_map_with_name_ref hands the plugins a lazy view of the map: MinFilter takes elevation / is_valid on the device (no plane is
    fetched to the host), a user plugin that indexes the view gets host planes on demand -- both see the same map"""
    from elevation_mapping_cupy_amd.elevation_mapping import LazyPlanes
    from elevation_mapping_cupy_amd.plugins.min_filter import MinFilter
    C = 130
    hip, _ = make_pair(eo.DEFAULTS, C, "reference_fp16", weights)
    R, t = fx.POSES["rotated"]
    hip.update_map_with_kernel(fx.cloud(C, 9000, 2), [], R, t.copy(), 0.0, 0.0)
    fetched = []
    view = LazyPlanes(7, lambda k: (fetched.append(k), hip.get_layer_raw(k))[1], device_map=hip)
    mf = MinFilter(cell_n=C, dilation_size=1, iteration_n=30, emap=hip)
    on_device = mf(view, hip.layer_names, None, [])
    assert fetched == []                                         # nothing crossed PCIe but the result
    from_host = mf(hip.elevation_map, hip.layer_names, None, [])
    assert np.array_equal(on_device, from_host, equal_nan=True)
    assert np.array_equal(view[2], hip.get_layer_raw(2)) and fetched == [2]      # a user plugin indexing the view
    assert np.asarray(view).shape == (7, C, C)
