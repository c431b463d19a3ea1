"""GPU: the call patterns of the reference's OWN test files, replayed through the reference's package name (``compat/`` on the path:
``from elevation_mapping_cupy import parameter, elevation_mapping``) with NumPy arrays where they build CuPy arrays --
EM/tests/test_elevation_mapping.py (the ``TestElevationMap`` class, all six fixture parametrisations), test_parameter.py,
test_semantic_map.py (``get_indices_fusion``) and test_plugins.py (the plugin manager on the test plugin configuration).
Those tests assert almost nothing (they are smoke tests of the API surface; two of them are stale against the reference's own code:
``get_fusion_of_pcl`` no longer exists there, ``exists_layer`` of a never-fused channel is False there too); here every call must
run, and what the reference's current code guarantees is asserted."""
import os
import pickle
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "compat"))

PLUGIN_YAML = """
min_filter: {enable: True, fill_nan: False, is_height_layer: True, layer_name: "min_filter", extra_params: {dilation_size: 1, iteration_n: 30}}
smooth_filter: {enable: True, fill_nan: False, is_height_layer: True, layer_name: "smooth", extra_params: {input_layer_name: "min_filter"}}
inpainting: {enable: True, fill_nan: False, is_height_layer: True, layer_name: "inpaint", extra_params: {method: "telea"}}
smooth_filter_1: {type: "smooth_filter", enable: True, fill_nan: False, is_height_layer: True, layer_name: "smooth_1", extra_params: {input_layer_name: "inpaint"}}
robot_centric_elevation: {enable: True, fill_nan: False, is_height_layer: True, layer_name: "robot_centric_elevation", extra_params: {resolution: 0.04, threshold: 1.1, use_threshold: True}}
semantic_filter: {type: "semantic_filter", enable: True, fill_nan: False, is_height_layer: False, layer_name: "sem_fil", extra_params: {classes: ['grass', 'tree', 'fence', 'person']}}
semantic_traversability: {type: "semantic_traversability", enable: True, fill_nan: False, is_height_layer: False, layer_name: "sem_traversability",
  extra_params: {layers: ['traversability', 'robot_centric_elevation'], thresholds: [0.7, 0.5], type: ['traversability', 'elevation']}}
"""


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    """config/weights.dat (the reference's pickle format, written from the committed golden weights) and the test plugin configuration"""
    d = tmp_path_factory.mktemp("refcfg")
    w = np.load(os.path.join(ROOT, "tests", "golden", "weights.npz"))
    with open(d / "weights.dat", "wb") as f:
        pickle.dump({"conv1.weight": w["w1"], "conv2.weight": w["w2"], "conv3.weight": w["w3"], "conv_final.weight": w["w_out"]}, f)
    (d / "plugin_config.yaml").write_text(PLUGIN_YAML)
    return str(d / "weights.dat"), str(d / "plugin_config.yaml")


def test_parameter(files):                                   # EM/tests/test_parameter.py
    from elevation_mapping_cupy.parameter import Parameter
    param = Parameter(use_chainer=False, weight_file=files[0], plugin_config_file=files[1])
    param.set_value("resolution", 0.1)
    assert len(param.get_types()) == len(param.get_names()) > 50
    assert {"float", "bool", "int", "str"} <= set(param.get_types())
    param.update()
    assert param.resolution == param.get_value("resolution") == 0.1 and param.cell_n == 82
    param.load_weights(param.weight_file)
    assert param.w1.shape == (4, 1, 3, 3) and param.w_out.shape == (1, 12, 1, 1)


CASES = [(["feat_0", "feat_1", "rgb"], ["average", "average", "color"]), (["feat_0", "feat_1"], ["average", "average"]),
         (["feat_0", "feat_1"], ["class_average", "class_average"]), (["feat_0", "feat_1"], ["class_bayesian", "class_bayesian"]),
         (["feat_0", "feat_1"], ["class_bayesian", "class_max"]), (["feat_0", "feat_1"], ["bayesian_inference", "bayesian_inference"])]


@pytest.mark.parametrize("add_lay,fusion_alg", CASES)
def test_elevation_map_suite(add_lay, fusion_alg, files):     # EM/tests/test_elevation_mapping.py: TestElevationMap, every method in file order
    from elevation_mapping_cupy import parameter, elevation_mapping
    p = parameter.Parameter(use_chainer=False, weight_file=files[0], plugin_config_file=files[1])
    p.subscriber_cfg["front_cam"]["channels"] = add_lay
    p.subscriber_cfg["front_cam"]["fusion"] = fusion_alg
    p.update()
    e = elevation_mapping.ElevationMap(p)
    rng = np.random.default_rng(11)
    # test_init
    assert len(e.layer_names) == e.elevation_map.shape[0] == 7
    # test_input: 100 000 points uniform in [0, 1), a random 3 x 3 matrix as "rotation" (not orthonormal), random t
    channels = ["x", "y", "z"] + e.param.additional_layers
    points = rng.random((100000, len(channels)), dtype=np.float32)
    R = rng.random((3, 3), dtype=np.float32); t = rng.random(3, dtype=np.float32)
    e.input_pointcloud(points, channels, R, t, 0, 0)
    assert (e.elevation_map[2] > 0.5).sum() > 100                     # points were fused
    assert e.exists_layer(e.param.additional_layers[0])               # the cloud's extra channel got its layer
    # test_update_normal
    before = e.elevation_map[3].copy()
    e.update_normal(e.elevation_map[0])
    nm = e.normal_map
    valid = e.elevation_map[2] > 0.5
    assert np.isfinite(nm).all() and (np.abs(np.linalg.norm(nm, axis=0) - 1)[valid][nm[2][valid] != 0] < 1e-5).all()
    assert np.array_equal(e.elevation_map[3], before)                 # update_normal leaves the traversability layer alone
    # test_move_to
    for i in range(20):
        e.move_to(np.array([i * 0.01, i * 0.02, i * 0.01]), rng.random((3, 3)))
    # test_get_map
    data = np.zeros((e.cell_n - 2, e.cell_n - 2), dtype=np.float32)
    for layer in ["elevation", "variance", "traversability", "min_filter", "smooth", "inpaint", "rgb"]:
        e.get_map_with_name_ref(layer, data)
    # test_get_position
    pos = rng.random((1, 3))
    e.get_position(pos)
    assert np.allclose(pos[0], e.center)
    # test_move
    e.move(rng.random(3))
    # test_exists_layer (what the reference's code guarantees: core layers, plugin layers, fused channels)
    for layer in ["elevation", "min_filter", "sem_fil", e.param.additional_layers[0]]:
        assert e.exists_layer(layer)
    # test_polygon_traversability
    result = np.array([0, 0, 0], np.float64)
    n = e.get_polygon_traversability(np.array([[0, 0], [2, 0], [0, 2]], dtype=np.float64), result)
    e.get_untraversable_polygon(np.zeros((n, 2)))
    # test_initialize_map
    for method in ["linear", "cubic", "nearest"]:
        e.initialize_map(np.array([[-4.0, 0.0, 0.0], [-4.0, 8.0, 1.0], [4.0, 8.0, 0.0], [4.0, 0.0, 0.0]]), method)
    # test_plugins: every configured plugin layer can be published
    data = np.zeros((200, 200), dtype=np.float32)
    assert len(e.plugin_manager.layer_names) == 7
    for layer in e.plugin_manager.layer_names:
        e.get_map_with_name_ref(layer, data)
    # test_clear
    e.clear()
    assert not e.elevation_map[2].any()


def test_class_max_cloud_of_the_reference_test(files):
    """the ``class_max`` branch of test_input (EM/tests/test_elevation_mapping.py:6-16,57-61: float16 probabilities and ids 0 / 1 packed
    by ``encode_max``) -- unreachable in the reference's own test ("class_max" is never IN the list of ``pointcloud_*`` module names),
    run here with the channel mapped to the fusion"""
    from elevation_mapping_cupy import parameter, elevation_mapping
    from elevation_mapping_cupy_amd.fusion.pointcloud_class_max import encode_max
    p = parameter.Parameter(use_chainer=False, weight_file=files[0], plugin_config_file=files[1])
    p.pointcloud_channel_fusions = {"max.*": "class_max"}
    p.update()
    e = elevation_mapping.ElevationMap(p)
    rng = np.random.default_rng(12)
    xyz = rng.random((100000, 3), dtype=np.float32)
    val = rng.random((100000, 2), dtype=np.float32).astype(np.float16)
    ind = rng.integers(0, 2, (100000, 2)).astype(np.uint32)
    points = np.column_stack([xyz, encode_max(val, ind)]).astype(np.float32)
    e.input_pointcloud(points, ["x", "y", "z", "max1", "max2"], rng.random((3, 3), dtype=np.float32), rng.random(3, dtype=np.float32), 0, 0)
    sm = e.semantic_map.semantic_map
    tot = sm[0] + sm[1]
    assert e.semantic_map.layer_names == ["max1", "max2"] and (sm[0] > 0).sum() > 50
    assert np.all((np.abs(tot - 1) < 1e-6) | (tot == 0))
    assert set(np.unique(e.semantic_map.get_id_max("max1"))) <= {0, 1}


@pytest.mark.parametrize("channels", [["rgb"], ["rgb", "feat_0"], []])
def test_indices_fusion(channels, files):                      # EM/tests/test_semantic_map.py::test_indices_fusion
    from elevation_mapping_cupy import parameter, elevation_mapping
    p = parameter.Parameter(use_chainer=False, weight_file=files[0], plugin_config_file=files[1])
    p.pointcloud_channel_fusions = {"rgb": "color", "default": "average"}
    p.update()
    sm = elevation_mapping.ElevationMap(p).semantic_map
    chans, fusions = sm.prepare(channels)
    pcl_indices, layer_indices = sm.get_indices_fusion(pcl_channels=channels, fusion_alg="average", layer_specs=sm.layer_specs_points)
    assert len(pcl_indices) == len(layer_indices) == sum(c != "rgb" for c in channels)
    assert all(isinstance(f, str) for f in fusions) and len(fusions) == len(channels)


def test_plugin_manager_on_the_test_configuration(files):     # EM/tests/test_plugins.py::test_plugin_manager
    from elevation_mapping_cupy import parameter, elevation_mapping
    from elevation_mapping_cupy.plugins.plugin_manager import PluginManager
    p = parameter.Parameter(use_chainer=False, weight_file=files[0], plugin_config_file=files[1])
    p.pointcloud_channel_fusions = {"default": "class_average"}
    p.update()
    e = elevation_mapping.ElevationMap(p)
    for name in ("grass", "tree", "fence", "person"):
        e.semantic_map.add_layer(name)
    C = e.cell_n
    manager = PluginManager(C, emap=e)
    manager.load_plugin_settings(files[1])
    rng = np.random.default_rng(13)
    em = np.zeros((7, C, C), np.float32)
    em[0] = rng.normal(size=(C, C)); em[2] = np.abs(rng.normal(size=(C, C)))
    names = ["elevation", "variance", "is_valid", "traversability", "time", "upper_bound", "is_upper_bound"]
    sem, sem_names, rot = e.semantic_map.semantic_map, e.semantic_map.layer_names, np.eye(3, dtype=np.float32)
    manager.update_with_name("min_filter", em, names)
    manager.update_with_name("smooth_filter", em, names)
    manager.update_with_name("semantic_filter", em, names, sem, sem_names, rot)
    manager.update_with_name("semantic_traversability", em, names, sem, sem_names)
    assert manager.get_map_with_name("smooth").shape == (C, C)
    for lay in manager.get_layer_names():
        manager.update_with_name(lay, em, names, sem, sem_names, rot, e.semantic_map.elements_to_shift)
        assert np.isfinite(manager.get_map_with_name(lay)[em[2] > 0.5]).all() or lay in ("min_filter", "smooth", "inpaint", "smooth_1")
