"""MI355X-native elevation-map fusion core behind the reference's ElevationMap / Parameter API."""
from .parameter import Parameter  # noqa: F401


def __getattr__(name):
    if name == "ElevationMap":
        from .elevation_mapping import ElevationMap
        return ElevationMap
    raise AttributeError(name)
