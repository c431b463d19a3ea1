"""GPU: the nine kernel factories of ``elevation_mapping_cupy.kernels`` (reference EM/kernels/custom_semantic_kernels.py), imported
through the drop-in package ``compat/elevation_mapping_cupy`` and called with the reference's argument lists, against the outputs of
the reference's OWN kernel source on the same inputs (tests/golden/semantic_toy.npz, made by tests/golden/make_golden.py from the
kernels compiled for the host): the toy case of EM/tests/test_semantic_kernels.py:25-307 (which only prints) and a richer draw."""
import os
import sys

import numpy as np
import pytest

import _fixtures as fx

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Factories:
    """the call surface of oracle/ref_kernels.RefKernels on top of the compat package's factories"""

    def __init__(self):
        sys.path.insert(0, os.path.join(ROOT, "compat"))
        from elevation_mapping_cupy.kernels import (sum_kernel, sum_compact_kernel, sum_max_kernel, alpha_kernel, average_kernel,
                                                    bayesian_inference_kernel, class_average_kernel, add_color_kernel, color_average_kernel)
        self.k = dict(sem_sum=sum_kernel(0.9, 4, 4), sum_compact=sum_compact_kernel(0.9, 4, 4), sem_sum_max=sum_max_kernel(0.9, 4, 4),
                      alpha=alpha_kernel(0.9, 4, 4), sem_average=average_kernel(4, 4), bayesian_inference=bayesian_inference_kernel(4, 4),
                      sem_class_average=class_average_kernel(4, 4, 0.5), sem_add_color=add_color_kernel(4, 4),
                      sem_color_average=color_average_kernel(4, 4))

    def __getattr__(self, name):
        k = self.__dict__["k"][name]
        return lambda *a: k(*a[:-1], size=a[-1])


@pytest.mark.parametrize("case", ["toy", "rich"])
def test_factories_reproduce_the_reference_kernels(case):
    g = np.load(os.path.join(ROOT, "tests", "golden", "semantic_toy.npz"))
    got = fx.semantic_kernel_run(_Factories(), fx.semantic_kernel_cases()[case])
    for name, arr in got.items():
        want = g["%s_%s" % (case, name)]
        if arr.dtype == np.uint32 or "color" in name:
            assert np.array_equal(arr.view(np.uint32), want.view(np.uint32)), name             # integer colour arithmetic: exact
        else:
            assert np.allclose(arr, want, atol=1e-6, rtol=1e-6), "%s: max |d| = %g" % (name, np.abs(arr - want).max())   # float atomics: order


def test_factory_arguments_are_checked():
    from elevation_mapping_cupy_amd._lib import EmapError
    K = _Factories()
    c = fx.semantic_kernel_cases()["toy"]
    with pytest.raises(EmapError):      # more elements than points x channels
        K.sem_sum(c["points"], np.eye(3, dtype=np.float32), np.zeros(3, np.float32), c["pcl_ids"], c["layer_ids"], np.array([6, 2], np.int32),
                  np.zeros((4, 4, 4), np.float32), np.zeros((4, 4, 4), np.float32), 1000)


def test_index_contents_are_checked_before_anything_is_launched():
    """the layer / channel / class-id INDEX ARRAYS decide which plane a thread writes: values outside the caller's arrays must be
    refused on the host (a bad index used to corrupt device memory of the shared scratch context silently), and a
    cell index outside the planes is dropped by the kernel"""
    from elevation_mapping_cupy_amd._lib import EmapError
    K = _Factories()
    c = fx.semantic_kernel_cases()["rich"]
    R, t = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    pts, chn = c["points"], np.array([6, 2, 2], np.int32)
    n = pts.shape[0]
    with pytest.raises(EmapError, match="map layer index"):          # layer 4 of a 4-plane newmap
        K.sem_sum(pts, R, t, c["pcl_ids"], np.array([1, 4], np.int32), chn, np.zeros((4, 4, 4), np.float32), np.zeros((4, 4, 4), np.float32), n * 2)
    with pytest.raises(EmapError, match="pcl channel index"):        # column 6 of 6-float rows
        K.sem_sum(pts, R, t, np.array([3, 6], np.int32), c["layer_ids"], chn, np.zeros((4, 4, 4), np.float32), np.zeros((4, 4, 4), np.float32), n * 2)
    with pytest.raises(EmapError, match="class id"):
        bad = c["max_id"].copy(); bad[3, 1] = 4
        K.sem_sum_max(pts, c["max_pt"], bad, c["pcl_ids"], c["layer_ids"], chn, np.zeros((4, 4, 4), np.float32), n)
    with pytest.raises(EmapError, match="3 n_ch \\+ 1"):             # two colour channels need 7 planes
        K.sem_add_color(c["points_color"], R, t, np.array([3, 4], np.int32), np.array([0, 1], np.int32), np.array([6, 2], np.int32), np.zeros((4, 4, 4), np.uint32), n)
    with pytest.raises(EmapError, match="map layer index"):
        K.sem_average(np.zeros((4, 4, 4), np.float32), c["pcl_ids"], np.array([1, -1], np.int32), chn, c["new_elmap"], np.zeros((4, 4, 4), np.float32), 32)
    with pytest.raises(EmapError, match="3 n_ch \\+ 1"):
        K.sem_color_average(np.zeros((4, 4, 4), np.uint32), np.array([3, 4], np.int32), np.array([0, 1], np.int32), np.array([6, 2], np.int32), np.zeros((4, 4, 4), np.float32), 32)
    # a point whose cell index lies outside the 4 x 4 planes: dropped, the planes next to it stay untouched
    far = pts.copy(); far[:, 0] = 16 + np.arange(n)
    newmap = np.zeros((4, 4, 4), np.float32)
    K.sem_sum(far, R, t, c["pcl_ids"], c["layer_ids"], chn, np.zeros((4, 4, 4), np.float32), newmap, n * 2)
    assert not newmap.any()
