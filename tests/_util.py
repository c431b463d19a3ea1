"""Test helpers: build matching (HIP ElevationMap, OracleMap) pairs from one config dict."""
import numpy as np

from oracle import emap_oracle as eo

PLANE_NAMES = ["elevation", "variance", "is_valid", "traversability", "time", "upper_bound", "is_upper_bound"]


def make_parameter(cfg, cell_n, mode="reference_fp16", weights=None):
    from elevation_mapping_cupy_amd.configs import parameter_from
    full = dict(eo.DEFAULTS)
    full.update(cfg)
    return parameter_from(full, cell_n, mode, weights)


def make_pair(cfg, cell_n, mode="reference_fp16", weights=None):
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    hip = ElevationMap(make_parameter(cfg, cell_n, mode, weights))
    orc = eo.OracleMap(eo.make_params(cfg, cell_n=cell_n, mode=mode, weights=weights))
    return hip, orc


def assert_planes_close(a, b, atol=1e-5, rtol=1e-5, names=PLANE_NAMES, max_bad=0, what=""):
    """fused height/variance within 1e-5 (north_star); flags exact."""
    for k in range(a.shape[0]):
        x, y = a[k].astype(np.float64), b[k].astype(np.float64)
        bad = ~(np.abs(x - y) <= atol + rtol * np.abs(y))
        bad &= ~(np.isnan(x) & np.isnan(y))
        n = int(bad.sum())
        name = names[k] if k < len(names) else str(k)
        assert n <= max_bad, "%s plane %s: %d cells differ, max |d| = %g (first at %s: %r vs %r)" % (
            what, name, n, np.nanmax(np.abs(x - y)), tuple(np.argwhere(bad)[0]), x[bad][0], y[bad][0])


def assert_planes_equal(a, b, names=PLANE_NAMES, what=""):
    """BIT-FOR-BIT equality of whole planes.  The oracle accumulates the same integers / fixed-point sums as the HIP kernels
    (oracle/emap_oracle.c) and both use the same correctly rounded float operations, so any difference is a defect, not rounding."""
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, (a.shape, b.shape)
    for k in range(a.shape[0]):
        x, y = a[k].view(np.uint32), b[k].view(np.uint32)
        if not np.array_equal(x, y):
            bad = x != y
            name = names[k] if k < len(names) else str(k)
            i = tuple(np.argwhere(bad)[0])
            raise AssertionError("%s plane %s: %d cells differ bitwise, max |d| = %g (first at %s: %r vs %r)" % (
                what, name, int(bad.sum()), float(np.nanmax(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)))), i, a[k][i], b[k][i]))
