"""CPU: the Navier-Stokes based fast-marching fill of the Inpainting plugin, method "ns" (host code of libemap_hip.so, as in the reference
where cv2.inpaint(..., cv2.INPAINT_NS) runs on the CPU -- EM/plugins/inpainting.py:33-38,59).  OpenCV is absent: parity with its values
is NOT pinned; pinned here: the C++ against a line-by-line Python restatement (oracle/ns_inpaint.py), and what every correct
implementation must do."""
import ctypes as ct

import numpy as np
import pytest

from elevation_mapping_cupy_amd import _lib
from oracle import ns_inpaint
from test_inpaint_telea import _case


def _c(image, mask, radius=1):
    lib = _lib.load()
    image = np.ascontiguousarray(image, np.uint8); mask = np.ascontiguousarray(mask, np.uint8)
    out = np.empty_like(image)
    p = lambda a: a.ctypes.data_as(ct.POINTER(ct.c_uint8))      # noqa: E731
    rc = lib.emap_inpaint_ns_u8(p(image), p(mask), image.shape[0], image.shape[1], radius, p(out))
    assert rc == 0
    return out


@pytest.mark.parametrize("seed,radius", [(1, 1), (2, 1), (3, 2), (4, 3), (5, 5)])
def test_cpp_equals_python_restatement(seed, radius):
    img, mask = _case(seed)
    got, want = _c(img, mask, radius), ns_inpaint.inpaint_ns(img, mask, radius)
    assert np.array_equal(got, want), "%d pixels differ" % int((got != want).sum())


def test_properties():
    img, mask = _case(7, n=40)
    out = _c(img, mask)
    assert np.array_equal(out[mask == 0], img[mask == 0])                   # known pixels are never touched
    known = img[mask == 0]
    assert out[mask != 0].min() >= int(known.min()) and out[mask != 0].max() <= int(known.max())     # a weighted MEAN of known pixels: no overshoot
    flat = np.full((20, 20), 117, np.uint8); m = np.zeros((20, 20), np.uint8); m[5:15, 6:13] = 1
    assert np.array_equal(_c(flat * (m == 0), m), flat)                     # a constant image is reproduced
    # level lines are continued: an image that only varies along x, with a hole -- every filled pixel lies between the known values
    # left and right of the hole, and rows far from the hole's upper / lower edge are filled alike
    y, x = np.mgrid[0:32, 0:32]
    ramp = (30 + 5 * x).astype(np.uint8); m = np.zeros((32, 32), np.uint8); m[6:26, 12:20] = 1
    filled = _c(np.where(m == 0, ramp, 0).astype(np.uint8), m, 2)
    inside = filled[6:26, 12:20].astype(int)
    assert inside.min() >= int(ramp[0, 10]) and inside.max() <= int(ramp[0, 21])
    assert np.abs(inside[8:12] - inside[9:13]).max() <= 2
    assert np.array_equal(_c(img, np.zeros_like(mask)), img)                # nothing to fill
    assert _lib.load().emap_inpaint_ns_u8(None, None, 4, 4, 1, None) != 0
    one = np.zeros((1, 5), np.uint8)            # a single row / column: the neighbour stencil would leave the image
    assert _lib.load().emap_inpaint_ns_u8(one.ctypes.data_as(ct.c_void_p), one.ctypes.data_as(ct.c_void_p), 1, 5, 1, one.ctypes.data_as(ct.c_void_p)) != 0
    assert _lib.load().emap_inpaint_ns_u8(one.ctypes.data_as(ct.c_void_p), one.ctypes.data_as(ct.c_void_p), 5, 1, 1, one.ctypes.data_as(ct.c_void_p)) != 0


def test_differs_from_telea_where_it_should():
    """the two methods share the march but not the estimate: on a textured image they must not be the same function"""
    img, mask = _case(11, n=36)
    lib = _lib.load()
    p = lambda a: a.ctypes.data_as(ct.POINTER(ct.c_uint8))      # noqa: E731
    t = np.empty_like(img)
    assert lib.emap_inpaint_telea_u8(p(np.ascontiguousarray(img)), p(np.ascontiguousarray(mask)), 36, 36, 2, p(t)) == 0
    n = _c(img, mask, 2)
    assert (t != n).sum() > 10 and np.abs(t.astype(int) - n.astype(int))[mask != 0].mean() < 30


def test_plugin_routes_method_ns():
    from elevation_mapping_cupy_amd.plugins.inpainting import Inpainting
    n = 30
    rng = np.random.default_rng(4)
    emap = np.zeros((7, n, n), np.float32)
    emap[0] = rng.uniform(-1, 2, (n, n)); emap[2] = rng.uniform(0, 1, (n, n)) < 0.8
    plug = Inpainting(cell_n=n, method="ns")
    assert plug.method == "ns" and Inpainting(cell_n=n, method="no such method").method == "telea"
    out = plug(emap, [], None, [])
    known = emap[2] >= 0.5
    hmin, hmax = float(emap[0][known].min()), float(emap[0][known].max())
    q = ((emap[0] - hmin) * 255 / (hmax - hmin)).astype(np.uint8)
    want = ns_inpaint.inpaint_ns(q, (~known).astype(np.uint8), 1).astype(np.float32) * (hmax - hmin) / 255 + hmin
    assert out.dtype == np.float64 and np.allclose(out, want, atol=1e-6)
