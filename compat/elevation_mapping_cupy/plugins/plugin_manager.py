from elevation_mapping_cupy_amd.plugins.plugin_manager import *  # noqa: F401,F403
from elevation_mapping_cupy_amd.plugins.plugin_manager import PluginBase, PluginManager, PluginParams  # noqa: F401
