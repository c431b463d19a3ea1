"""Import-compatible alias of the reference package name.  Put ``<repo>/compat`` (and ``<repo>``) on PYTHONPATH and the C++ ROS
wrapper's ``py::module::import("elevation_mapping_cupy.elevation_mapping")`` / ``...parameter`` (src/elevation_mapping_wrapper.cpp:34-38)
resolve to the MI355X implementation without touching the node."""
from elevation_mapping_cupy_amd.parameter import Parameter  # noqa: F401
from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap  # noqa: F401
