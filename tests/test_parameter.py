"""CPU: Parameter keeps the reference's reflection surface (EM/parameter.py; used by wrapper.cpp:45-77)."""
import numpy as np


def test_defaults_and_reflection(tmp_path):
    from elevation_mapping_cupy_amd import Parameter
    p = Parameter()
    names, types = p.get_names(), p.get_types()
    assert len(names) == len(types)
    d = dict(zip(names, types))
    assert d["resolution"] == "float" and d["enable_edge_sharpen"] == "bool" and d["checker_layer"] == "str" and d["max_unsafe_n"] == "int"
    p.set_value("resolution", 0.1); assert p.get_value("resolution") == 0.1
    p.set_value("resolution", 0.04); p.update()
    assert p.cell_n == 202 and p.true_cell_n == 200 and abs(p.true_map_length - 8.0) < 1e-9
    assert p.pointcloud_channel_fusions == {"rgb": "color", "default": "class_average"}
    assert p.initial_variance == 10.0 and p.max_ray_length == 2.0 and p.dilation_size == 2      # dataclass defaults
    import pickle
    w = {"conv1.weight": np.ones((4, 1, 3, 3), np.float32), "conv2.weight": np.zeros((4, 1, 3, 3), np.float32),
         "conv3.weight": np.zeros((4, 1, 3, 3), np.float32), "conv_final.weight": np.ones((1, 12, 1, 1), np.float32)}
    f = tmp_path / "w.dat"
    f.write_bytes(pickle.dumps(w))
    p.load_weights(str(f))
    assert p.w1.shape == (4, 1, 3, 3) and p.w_out.sum() == 12


def test_parameter_tables_agree_with_oracle_tables():
    from elevation_mapping_cupy_amd.configs import CORE_PARAM_YAML, parameter_from
    from oracle import emap_oracle as eo
    for k, v in CORE_PARAM_YAML.items():
        assert float(eo.YAML[k]) == float(v), k
    assert parameter_from(CORE_PARAM_YAML, 1024).cell_n == 1024


def test_reference_package_name_alias_imports():
    """compat/elevation_mapping_cupy re-exports the classes under the names the ROS wrapper imports (no GPU needed to import)"""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "compat"))
    try:
        em = importlib.import_module("elevation_mapping_cupy.elevation_mapping")
        pm = importlib.import_module("elevation_mapping_cupy.parameter")
        pl = importlib.import_module("elevation_mapping_cupy.plugins.plugin_manager")
        fu = importlib.import_module("elevation_mapping_cupy.fusion.fusion_manager")
        from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
        from elevation_mapping_cupy_amd.parameter import Parameter
        assert em.ElevationMap is ElevationMap and pm.Parameter is Parameter
        assert hasattr(pl, "PluginBase") and hasattr(fu, "FusionBase")
        p = pm.Parameter()
        assert "resolution" in p.get_names() and len(p.get_names()) == len(p.get_types())
    finally:
        sys.path.pop(0)
        for k in [k for k in sys.modules if k == "elevation_mapping_cupy" or k.startswith("elevation_mapping_cupy.")]:
            del sys.modules[k]
