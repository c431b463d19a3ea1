from elevation_mapping_cupy_amd.fusion.fusion_manager import *  # noqa: F401,F403
from elevation_mapping_cupy_amd.fusion.fusion_manager import FusionBase, FusionManager  # noqa: F401
