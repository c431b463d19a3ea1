"""``PluginBase`` / ``PluginParams`` / ``PluginManager`` -- the reference's plugin surface
(reference EM/plugins/plugin_manager.py:15-260): YAML keys ``enable, fill_nan, is_height_layer, layer_name,
extra_params[, type]``, discovery = first ``PluginBase`` subclass of module ``plugins.<type>``, call arity dispatch
(5/7/8/9 parameters).  Arrays are NumPy (host copies of device layers); the manager injects ``emap`` (the owning
``ElevationMap``) into plugins that accept it so that they can run their arithmetic on the device."""
from __future__ import annotations

import importlib
import sys
import inspect
from abc import ABC
from dataclasses import dataclass
from inspect import signature
from typing import Dict, List, Optional

import numpy as np
import yaml


@dataclass
class PluginParams:
    name: str
    layer_name: str
    fill_nan: bool = False        # fill nan to invalid region
    is_height_layer: bool = False  # if this is a height layer


class PluginBase(ABC):
    def __init__(self, *args, **kwargs):
        pass

    def __call__(self, elevation_map, layer_names, plugin_layers, plugin_layer_names, semantic_map, semantic_layer_names,
                 *args, **kwargs):
        """layers of elevation_map: 0 elevation, 1 variance, 2 is_valid, 3 traversability, 4 time, 5 upper_bound,
        6 is_upper_bound; return a (cell_n, cell_n) array."""
        pass

    def get_layer_data(self, elevation_map, layer_names, plugin_layers, plugin_layer_names, semantic_map,
                       semantic_layer_names, name: str) -> Optional[np.ndarray]:
        if name in layer_names:
            return elevation_map[layer_names.index(name)].copy()
        if name in plugin_layer_names:
            return plugin_layers[plugin_layer_names.index(name)].copy()
        if name in semantic_layer_names:
            return semantic_map[semantic_layer_names.index(name)].copy()
        print(f"Could not find layer {name}!")
        return None


class PluginManager(object):
    def __init__(self, cell_n: int, emap=None, package: str = "elevation_mapping_cupy_amd.plugins"):
        self.cell_n = cell_n
        self.emap = emap
        self.package = package
        self.plugin_params: List[PluginParams] = []
        self.plugins = []
        self.layers = np.zeros((0, cell_n, cell_n), np.float32)
        self.layer_names: List[str] = []
        self.plugin_names: List[str] = []

    def init(self, plugin_params: List[PluginParams], extra_params: List[Dict]):
        self.plugins = []
        kept = []
        for param, extra_param in zip(plugin_params, extra_params):
            try:
                m = importlib.import_module("." + param.name, package=self.package)
            except ImportError as ex:     # a configured plugin without an implementation here must not take the others down
                print("[WARNING] plugin {} is not available on this backend ({}); layer {} skipped".format(param.name, ex, param.layer_name),
                      file=sys.stderr)
                continue
            kept.append(param)
            for name, obj in inspect.getmembers(m):
                if inspect.isclass(obj) and issubclass(obj, PluginBase) and name != "PluginBase":
                    extra_param = dict(extra_param or {})
                    extra_param["cell_n"] = self.cell_n
                    if "emap" in signature(obj.__init__).parameters:
                        extra_param["emap"] = self.emap
                    self.plugins.append(obj(**extra_param))
        self.plugin_params = kept
        self.layers = np.zeros((len(self.plugins), self.cell_n, self.cell_n), dtype=np.float32)
        self.layer_names = [p.layer_name for p in self.plugin_params]
        self.plugin_names = [p.name for p in self.plugin_params]

    def load_plugin_settings(self, file_path: str):
        print("Start loading plugins...")
        with open(file_path, "r") as f:
            cfg = yaml.safe_load(f) or {}
        plugin_params, extra_params = [], []
        for k, v in cfg.items():
            if v["enable"]:
                plugin_params.append(PluginParams(name=k if "type" not in v else v["type"], layer_name=v["layer_name"],
                                                  fill_nan=v["fill_nan"], is_height_layer=v["is_height_layer"]))
                extra_params.append(v.get("extra_params", {}))
        self.init(plugin_params, extra_params)
        print("Loaded plugins are ", *self.plugin_names)

    def get_layer_names(self):
        return [p.layer_name for p in self.plugin_params]

    def get_plugin_names(self):
        return [p.name for p in self.plugin_params]

    def get_plugin_index_with_name(self, name: str):
        return self.plugin_names.index(name) if name in self.plugin_names else None

    def get_layer_index_with_name(self, name: str):
        return self.layer_names.index(name) if name in self.layer_names else None

    def update_with_name(self, name, elevation_map, layer_names, semantic_map=None, semantic_params=None, rotation=None,
                         elements_to_shift={}):
        idx = self.get_layer_index_with_name(name)
        if idx is None or idx >= len(self.plugins):
            return
        plugin = self.plugins[idx]
        n_param = len(signature(plugin).parameters)        # arity dispatch of the reference (:199-225)
        args = [elevation_map, layer_names, self.layers, self.layer_names]
        if n_param == 5:
            pass
        elif n_param == 7:
            args += [semantic_map, semantic_params]
        elif n_param == 8:
            args += [semantic_map, semantic_params, rotation]
        else:
            args += [semantic_map, semantic_params, rotation, elements_to_shift]
        out = plugin(*args)
        if out is None:                    # (semantic_traversability without its input layer: the reference prints and returns nothing)
            return
        self.layers[idx] = np.asarray(out, np.float32)

    def get_map_with_name(self, name: str):
        idx = self.get_layer_index_with_name(name)
        return None if idx is None else self.layers[idx]

    def get_param_with_name(self, name: str):
        idx = self.get_layer_index_with_name(name)
        return None if idx is None else self.plugin_params[idx]
