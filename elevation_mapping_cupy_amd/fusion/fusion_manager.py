"""``FusionBase`` / ``FusionManager`` -- same registration-by-module-name surface as the reference
(reference EM/fusion/fusion_manager.py:18-80).  The reference plugin objects launch their own CuPy kernels; here a
plugin only declares which device-side fusion it stands for (``kind``) and ``SemanticMap`` collects them into ONE
``emap_sem_spec`` per frame, so the cloud is read once for all channels."""
from __future__ import annotations

import importlib
import inspect
import sys
from abc import ABC, abstractmethod
from typing import Dict


class FusionBase(ABC):
    @abstractmethod
    def __init__(self, *args, **kwargs):
        self.name = None
        self.kind = None  # "average" | "class_average" | "class_bayesian" | "bayesian_inference" | "color"


class FusionManager(object):
    def __init__(self, params):
        self.fusion_plugins: Dict[str, FusionBase] = {}
        self.params = params
        self.plugins = []
        self.unavailable = []

    def register_plugin(self, plugin):
        """``plugin`` = module name under ``elevation_mapping_cupy_amd.fusion`` (e.g. ``pointcloud_average``).
        A module that does not exist here is reported when a channel asks for it, and skipped."""
        try:
            m = importlib.import_module("." + plugin, package="elevation_mapping_cupy_amd.fusion")
        except ImportError:
            self.unavailable.append(plugin)   # reported only if a channel actually asks for it (get_plugin_idx)
            return False
        for name, obj in inspect.getmembers(m):
            if inspect.isclass(obj) and issubclass(obj, FusionBase) and name != "FusionBase":
                self.plugins.append(obj(self.params))
        return True

    def get_plugin_idx(self, name: str, data_type: str):
        name = data_type + "_" + name
        for idx, plugin in enumerate(self.plugins):
            if plugin.name == name:
                return idx
        print("[WARNING] Plugin {} is not in the list: {} (not available on the MI355X backend: {})".format(
            name, [p.name for p in self.plugins], self.unavailable), file=sys.stderr)
        return None

    def get_plugin(self, name: str, data_type: str = "pointcloud"):
        idx = self.get_plugin_idx(name, data_type)
        return None if idx is None else self.plugins[idx]
