"""CPU: the publish-time plugins that are array glue in the reference (EM/plugins/{robot_centric_elevation, semantic_filter,
semantic_traversability, max_layer_filter, features_pca}.py) -- NumPy here, no device needed; semantics checked on hand-sized inputs,
the colour table against the PASCAL-VOC bit rule, the PCA against scikit-learn, the call-arity dispatch through the PluginManager."""
import numpy as np

from elevation_mapping_cupy_amd.plugins.plugin_manager import PluginManager, PluginParams

NAMES = ["elevation", "variance", "is_valid", "traversability", "time", "upper_bound", "is_upper_bound"]


def _map(C=6, seed=0):
    rng = np.random.default_rng(seed)
    m = np.zeros((7, C, C), np.float32)
    m[0] = rng.normal(0, 0.3, (C, C)); m[2] = (rng.random((C, C)) < 0.7); m[3] = rng.random((C, C))
    return m


def test_robot_centric_elevation():
    from elevation_mapping_cupy_amd.plugins.robot_centric_elevation import RobotCentricElevation
    C, res = 6, 0.04
    m = _map(C)
    R = np.array([[1, 0, 0], [0, 0.8, -0.6], [0.1, 0.6, 0.8]], np.float32)
    out = RobotCentricElevation(cell_n=C, resolution=res)(m, NAMES, None, [], None, [], R)
    for i in range(C):
        for j in range(C):
            want = np.float32(R[2, 0] * np.float32(i * res) + R[2, 1] * np.float32(j * res) + R[2, 2] * m[0, i, j]) if m[2, i, j] > 0.5 else m[0, i, j]
            assert abs(out[i, j] - want) < 1e-6
    thr = RobotCentricElevation(cell_n=C, resolution=res, threshold=0.05, use_threshold=True)(m, NAMES, None, [], None, [], R)
    valid = m[2] > 0.5
    assert set(np.unique(thr[valid])) <= {0.0, 1.0} and np.array_equal(thr[~valid], m[0][~valid])


def test_semantic_filter_colour_table_and_argmax():
    from elevation_mapping_cupy_amd.plugins.semantic_filter import SemanticFilter, _voc_colors
    table = _voc_colors(255)

    def voc(i):                                    # the PASCAL-VOC rule: bit k of the class index goes to channel k mod 3, from the top bit down
        r = g = b = 0
        for j in range(8):
            r |= ((i >> 0) & 1) << (7 - j); g |= ((i >> 1) & 1) << (7 - j); b |= ((i >> 2) & 1) << (7 - j)
            i >>= 3
        return r, g, b
    for i in (4, 5, 8, 21, 100, 255):
        assert tuple(table[i - 1]) == voc(i)
    assert tuple(table[0]) == (81, 113, 162) and tuple(table[1]) == (81, 113, 162) and tuple(table[2]) == (188, 63, 59)
    C = 5
    sem = np.random.default_rng(1).random((3, C, C)).astype(np.float32)
    f = SemanticFilter(cell_n=C, classes=["grass", "tre.*"])
    out = f(_map(C), NAMES, np.zeros((0, C, C), np.float32), [], sem, ["grass", "person", "tree"], np.eye(3), {})
    want_id = np.argmax(sem[[0, 2]], axis=0)
    assert np.array_equal(out.view(np.uint32), f.color_encoding.view(np.uint32)[want_id])
    none = f(_map(C), NAMES, np.zeros((0, C, C), np.float32), [], sem, ["a", "b", "c"], np.eye(3), {})
    assert np.all(none.view(np.uint32) == f.color_encoding.view(np.uint32)[0])


def test_semantic_traversability_votes():
    from elevation_mapping_cupy_amd.plugins.semantic_traversability import SemanticTraversability
    C = 6
    m = _map(C)
    rce = np.random.default_rng(2).normal(0.5, 0.3, (1, C, C)).astype(np.float32)
    p = SemanticTraversability(cell_n=C, layers=["traversability", "robot_centric_elevation"], thresholds=[0.7, 0.5], type=["traversability", "elevation"])
    out = p(m, NAMES, rce, ["robot_centric_elevation"], None, [])
    want = np.where(((m[3] <= 0.7).astype(int) + (rce[0] >= 0.5).astype(int)) >= 1, 1.0, 0.1)
    assert np.allclose(out, want)
    assert p(m, NAMES, rce, ["other"], None, []) is None


def test_max_layer_filter():
    from elevation_mapping_cupy_amd.plugins.max_layer_filter import MaxLayerFilter
    C = 5
    m = _map(C)
    sem = np.random.default_rng(3).random((2, C, C)).astype(np.float32); sem[0][0, 0] = 0.0
    p = MaxLayerFilter(cell_n=C, layers=["traversability", "grass", "nope"], reverse=[True, False, False], thresholds=[False, 0.4, False],
                       scales=[2.0, 1.0, 1.0], default_value=0.25)
    out = p(m, NAMES, np.zeros((0, C, C), np.float32), [], sem, ["grass", "tree"])
    a = (1.0 - np.where(m[3] == 0, 0.25, m[3])) * 2.0
    b = (np.where(sem[0] == 0, 0.25, sem[0]) > 0.4).astype(float)
    assert np.allclose(out, np.maximum(a, b))
    lo = MaxLayerFilter(cell_n=C, layers=["traversability", "grass"], reverse=[False, False], min_or_max="min", thresholds=[False, False])(
        m, NAMES, np.zeros((0, C, C), np.float32), [], sem, ["grass", "tree"])
    assert np.allclose(lo, np.minimum(np.where(m[3] == 0, 0.0, m[3]), sem[0]))
    empty = MaxLayerFilter(cell_n=C, layers=["x"], default_value=0.5)(m, NAMES, np.zeros((0, C, C), np.float32), [], sem, ["grass", "tree"])
    assert np.all(empty == 0.5)


def test_features_pca_against_scikit_learn():
    from sklearn.decomposition import PCA
    from elevation_mapping_cupy_amd.plugins.features_pca import FeaturesPca, _pca3
    rng = np.random.default_rng(4)
    data = rng.normal(0, 1, (400, 6)) @ rng.normal(0, 1, (6, 6))
    mine, ref = _pca3(data), PCA(n_components=3).fit(data).transform(data)
    assert np.allclose(np.abs(mine), np.abs(ref), atol=1e-9)              # same axes; the sign convention is scikit-learn's own and version dependent
    C = 12
    sem = rng.normal(0, 0.6, (5, C, C)).astype(np.float32)
    out = FeaturesPca(cell_n=C, process_layer_names=["feat_.*"])(_map(C), NAMES, np.zeros((0, C, C), np.float32), [], sem,
                                                                   ["feat_0", "feat_1", "rgb", "feat_2", "feat_3"])
    px = out.view(np.uint32)
    assert out.shape == (C, C) and px.max() < (1 << 24) and len(np.unique(px)) > C            # a packed colour per cell, not constant
    for sh in (16, 8, 0):                                                                      # every component spans its full 0..255 range
        ch = (px >> sh) & 0xFF
        assert ch.min() == 0 and ch.max() == 255
    assert np.all(FeaturesPca(cell_n=C, process_layer_names=["zzz"])(_map(C), NAMES, np.zeros((0, C, C), np.float32), [], sem, ["a"] * 5) == 0)


def test_manager_dispatches_by_call_arity(tmp_path):
    """the reference's test plugin configuration (EM/tests/plugin_config.yaml: robot_centric_elevation, semantic_filter,
    semantic_traversability next to the filters) loads and every host plugin is called with the argument list its signature asks for"""
    cfg = tmp_path / "plugins.yaml"
    cfg.write_text("""
robot_centric_elevation:
  enable: True
  fill_nan: False
  is_height_layer: True
  layer_name: "robot_centric_elevation"
  extra_params: {resolution: 0.04, threshold: 1.1, use_threshold: True}
semantic_filter:
  type: "semantic_filter"
  enable: True
  fill_nan: False
  is_height_layer: False
  layer_name: "sem_fil"
  extra_params: {classes: ['grass', 'tree', 'fence', 'person']}
semantic_traversability:
  type: "semantic_traversability"
  enable: True
  fill_nan: False
  is_height_layer: False
  layer_name: "sem_traversability"
  extra_params: {layers: ['traversability', 'robot_centric_elevation'], thresholds: [0.7, 0.5], type: ['traversability', 'elevation']}
max_layer_filter:
  enable: True
  fill_nan: False
  is_height_layer: False
  layer_name: "max_categories"
  extra_params: {layers: ['grass', 'tree'], reverse: [False, False], thresholds: [False, False]}
features_pca:
  enable: True
  fill_nan: False
  is_height_layer: False
  layer_name: "pca"
  extra_params: {process_layer_names: ['grass', 'tree', 'person']}
""")
    C = 16
    man = PluginManager(C)
    man.load_plugin_settings(str(cfg))
    assert man.get_layer_names() == ["robot_centric_elevation", "sem_fil", "sem_traversability", "max_categories", "pca"]
    m = _map(C, 5)
    sem = np.random.default_rng(6).random((3, C, C)).astype(np.float32)
    for name in man.get_layer_names():
        man.update_with_name(name, m, NAMES, sem, ["grass", "tree", "person"], np.eye(3, dtype=np.float32), {})
    assert set(np.unique(man.get_map_with_name("robot_centric_elevation")[m[2] > 0.5])) <= {0.0, 1.0}
    assert set(np.unique(man.get_map_with_name("sem_traversability"))) <= {np.float32(0.1), np.float32(1.0)}
    assert np.allclose(man.get_map_with_name("max_categories"), np.maximum(sem[0], sem[1]))
    assert len(np.unique(man.get_map_with_name("pca").view(np.uint32))) > C
    assert isinstance(man.get_param_with_name("pca"), PluginParams)
