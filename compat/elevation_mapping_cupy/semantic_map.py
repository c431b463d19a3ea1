from elevation_mapping_cupy_amd.semantic_map import *  # noqa: F401,F403
from elevation_mapping_cupy_amd.semantic_map import SemanticMap  # noqa: F401
