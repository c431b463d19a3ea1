from elevation_mapping_cupy_amd.elevation_mapping import *  # noqa: F401,F403
from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap  # noqa: F401
