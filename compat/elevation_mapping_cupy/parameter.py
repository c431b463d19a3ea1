from elevation_mapping_cupy_amd.parameter import *  # noqa: F401,F403
from elevation_mapping_cupy_amd.parameter import Parameter  # noqa: F401
