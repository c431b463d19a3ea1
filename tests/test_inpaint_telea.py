"""CPU: Telea's fast-marching inpainting of the Inpainting plugin (host code of libemap_hip.so, as in the reference where
cv2.inpaint runs on the CPU -- EM/plugins/inpainting.py:59).  OpenCV is absent: parity with its values is NOT pinned; pinned here:
the C++ against a line-by-line Python restatement (oracle/telea.py), and what every correct implementation must do."""
import ctypes as ct

import numpy as np
import pytest

from elevation_mapping_cupy_amd import _lib
from oracle import telea


def _c(image, mask, radius=1):
    lib = _lib.load()
    image = np.ascontiguousarray(image, np.uint8); mask = np.ascontiguousarray(mask, np.uint8)
    out = np.empty_like(image)
    p = lambda a: a.ctypes.data_as(ct.POINTER(ct.c_uint8))      # noqa: E731
    rc = lib.emap_inpaint_telea_u8(p(image), p(mask), image.shape[0], image.shape[1], radius, p(out))
    assert rc == 0
    return out


def _case(seed, n=28, frac=0.25):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:n, 0:n]
    img = np.clip(60 + 3 * x + 2 * y + 25 * np.sin(x / 4.0) + rng.normal(0, 2, (n, n)), 0, 255).astype(np.uint8)
    mask = (rng.uniform(0, 1, (n, n)) < frac).astype(np.uint8)
    mask[8:14, 10:19] = 1                                    # a hole several pixels deep: the front really marches
    mask[0, :5] = 1; mask[-1, -4:] = 1; mask[5:9, 0] = 1       # holes touching the image border
    return img, mask


@pytest.mark.parametrize("seed,radius", [(1, 1), (2, 1), (3, 2), (4, 3)])
def test_cpp_equals_python_restatement(seed, radius):
    img, mask = _case(seed)
    got, want = _c(img, mask, radius), telea.inpaint_telea(img, mask, radius)
    assert np.array_equal(got, want), "%d pixels differ" % int((got != want).sum())


def test_properties():
    img, mask = _case(7, n=40)
    out = _c(img, mask)
    assert np.array_equal(out[mask == 0], img[mask == 0])                   # known pixels are never touched
    known = img[mask == 0]
    assert out[mask != 0].min() >= int(known.min()) - 40 and out[mask != 0].max() <= int(known.max()) + 40
    flat = np.full((20, 20), 117, np.uint8); m = np.zeros((20, 20), np.uint8); m[5:15, 6:13] = 1
    assert np.array_equal(_c(flat * (m == 0), m), flat)                     # a constant image is reproduced
    y, x = np.mgrid[0:24, 0:24]
    ramp = (40 + 4 * x).astype(np.uint8); m = np.zeros((24, 24), np.uint8); m[8:16, 8:16] = 1
    filled = _c(np.where(m == 0, ramp, 0).astype(np.uint8), m)
    err = np.abs(filled.astype(int) - ramp.astype(int))[m != 0]
    assert err.max() <= 12 and err.mean() <= 5                               # radius 1 and a unit-length gradient term: a ramp of 32 levels across the hole comes back within a third
    assert np.array_equal(_c(img, np.zeros_like(mask)), img)                # nothing to fill
    assert _lib.load().emap_inpaint_telea_u8(None, None, 4, 4, 1, None) != 0
    lib = _lib.load()                                                       # one row / one column: rejected (the gradient term's neighbours would leave the image)
    for shape in ((1, 9), (9, 1), (1, 1)):
        a, m1, o = np.full(shape, 7, np.uint8), np.ones(shape, np.uint8), np.zeros(shape, np.uint8)
        assert lib.emap_inpaint_telea_u8(a.ctypes.data_as(ct.POINTER(ct.c_uint8)), m1.ctypes.data_as(ct.POINTER(ct.c_uint8)), shape[0], shape[1], 1, o.ctypes.data_as(ct.POINTER(ct.c_uint8))) != 0


def test_plugin_routes_methods():
    """method 'telea' (the reference's default) runs the fast-marching fill on the host; the quantisation around it is the reference's"""
    from elevation_mapping_cupy_amd.plugins.inpainting import Inpainting
    n = 30
    rng = np.random.default_rng(3)
    emap = np.zeros((7, n, n), np.float32)
    emap[0] = rng.uniform(-1, 2, (n, n)); emap[2] = rng.uniform(0, 1, (n, n)) < 0.8
    plug = Inpainting(cell_n=n, method="telea")
    out = plug(emap, [], None, [])
    known = emap[2] >= 0.5
    hmin, hmax = float(emap[0][known].min()), float(emap[0][known].max())
    q = ((emap[0] - hmin) * 255 / (hmax - hmin)).astype(np.uint8)
    want = telea.inpaint_telea(q, (~known).astype(np.uint8), 1).astype(np.float32) * (hmax - hmin) / 255 + hmin
    assert out.dtype == np.float64 and np.allclose(out, want, atol=1e-6)
