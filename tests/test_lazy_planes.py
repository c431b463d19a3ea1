"""CPU: ``LazyPlanes`` -- what a plugin receives instead of the reference's (L, C, C) cupy array -- behaves like that array
(shape / dtype / ndim / indexing / arithmetic / ndarray methods) while fetching a plane only when it is touched."""
import numpy as np
import pytest

from elevation_mapping_cupy_amd.elevation_mapping import LazyPlanes


def _planes(log):
    def fetch(k):
        log.append(k)
        return np.full((4, 5), float(k), np.float32)
    return LazyPlanes(3, fetch, rows=4, cols=5)


def test_array_surface_without_fetching():
    log = []
    L = _planes(log)
    assert L.shape == (3, 4, 5) and L.shape[1:] == (4, 5) and L.ndim == 3 and L.dtype == np.float32 and len(L) == 3
    assert log == []                                    # nothing crossed PCIe yet
    assert L[1][0, 0] == 1.0 and L[-1][0, 0] == 2.0 and log == [1, 2]
    with pytest.raises(IndexError):
        L[3]


def test_third_party_plugin_idioms():
    L = _planes([])
    assert (L * 2)[2, 0, 0] == 4.0 and (1 + L)[0, 0, 0] == 1.0 and (-L)[1, 0, 0] == -1.0
    c = L.copy(); c[0] += 7                             # elevation_map.copy()
    assert c[0, 0, 0] == 7.0 and L[0][0, 0] == 0.0
    assert float(L.sum()) == 60.0 and L.mean(axis=(1, 2)).tolist() == [0.0, 1.0, 2.0]
    assert (L > 0.5).sum() == 40 and L[1:, 2].shape == (2, 5)
    assert np.where(L[2] > 1, L[0], L[1]).shape == (4, 5) and np.asarray(L).shape == (3, 4, 5)
    assert [p[0, 0] for p in L] == [0.0, 1.0, 2.0]
