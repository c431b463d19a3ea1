"""Factories of EM/kernels/custom_image_kernels.py with the reference's signatures, routed to the camera-path calls of
``libemap_hip.so`` (include/emap_hip.h): ``image_to_map_correspondence_kernel`` (:9-157 -> emap_image_correspondence on a scratch
context that holds the caller's elevation planes) and the three samplers ``average_`` / ``exponential_`` /
``color_correspondences_to_map_kernel`` (:160-271 -> emap_image_fuse_arrays, the same device kernels on caller arrays).

Every factory returns ``k(*arrays, size=n)`` on NumPy arrays; the out-parameters (``uv_correspondence``, ``valid_correspondence``,
``new_sem_map``) are written in place like a CuPy ``raw`` out-parameter.  Scalars the reference passes by value (``x1``, ``y1``,
``z1``, ``map_idx``, image sizes; EM/elevation_mapping.py:540-554, fusion/image_*.py) may be Python numbers or one-element arrays.
The per-frame path is ``ElevationMap.input_image``; these are the L2 surface of SURVEY section 8(b)."""
from __future__ import annotations

import ctypes as ct

import numpy as np

from elevation_mapping_cupy_amd._lib import f32p
from .custom_kernels import _scalar, _scratch

KIND = {"exponential": 0, "color": 1, "average": 2}


def image_to_map_correspondence_kernel(resolution, width, height, tolerance_z_collision):
    em = _scratch(width, height, resolution)
    em._chk(em._lib.emap_image_set_tolerance(em._ctx, ct.c_double(float(tolerance_z_collision))))
    C = int(width)

    def kernel(map_, x1, y1, z1, P, K, D, image_height, image_width, center, uv_correspondence, valid_correspondence, size=None):
        planes = np.asarray(map_, np.float32).reshape(-1, C, C)
        em.set_layer_raw("elevation", np.ascontiguousarray(planes[0]))          # the kernel reads planes 0 (height) and 2 (is_valid) only
        em.set_layer_raw("is_valid", np.ascontiguousarray(planes[2]))
        a = lambda v, n: np.ascontiguousarray(np.asarray(v, np.float32).reshape(-1)[:n])  # noqa: E731
        Pm, Km, Dm, cen = a(P, 12), a(K, 9), np.zeros(5, np.float32), a(center, 3)
        d = np.asarray(D, np.float32).reshape(-1)[:5]
        Dm[:d.size] = d
        em._chk(em._lib.emap_image_correspondence(em._ctx, ct.c_float(_scalar(x1)), ct.c_float(_scalar(y1)), ct.c_float(_scalar(z1)),
                                                  f32p(Pm), f32p(Km), f32p(Dm), ct.c_float(_scalar(image_height)),
                                                  ct.c_float(_scalar(image_width)), f32p(cen)))
        uv, va = np.empty((2, C, C), np.float32), np.empty((C, C), np.uint8)
        em._chk(em._lib.emap_image_get_correspondence(em._ctx, f32p(uv), va.ctypes.data_as(ct.POINTER(ct.c_uint8))))
        # the reference kernel returns early for cells it rejects and leaves their entries as they were: write only what it writes
        uvo, vao = np.asarray(uv_correspondence), np.asarray(valid_correspondence)
        wrote = (planes[2] == 1)
        # cells that passed the validity test but were rejected before the occlusion walk (behind the camera / outside the image)
        # are left untouched by the reference as well; the library reports them as (0, 0, invalid), which equals the zero-initialised
        # arrays of the reference's caller (EM/elevation_mapping.py:101-102) -- copy them only where something was computed
        wrote &= (uv[0] != 0) | (uv[1] != 0) | (va != 0)
        uvo.reshape(2, C, C)[:, wrote] = uv[:, wrote]
        vao.reshape(C, C)[wrote] = va[wrote].astype(vao.dtype)
    return kernel


def _sampler(kind, width, height, alpha=0.0):
    em = _scratch(width, height)
    C = int(width)

    def kernel(sem_map, map_idx, image, uv_correspondence, valid_correspondence, image_height, image_width, new_sem_map, size=None):
        layer = int(_scalar(map_idx))
        src = np.ascontiguousarray(np.asarray(sem_map, np.float32).reshape(-1, C, C)[layer])
        H, W = int(_scalar(image_height)), int(_scalar(image_width))
        img = np.ascontiguousarray(np.asarray(image, np.float32).reshape(-1, H, W))
        uv = np.ascontiguousarray(np.asarray(uv_correspondence, np.float32).reshape(2, C, C))
        va = np.ascontiguousarray(np.asarray(valid_correspondence).reshape(C, C) != 0, np.uint8)
        out = np.empty((C, C), np.float32)
        em._chk(em._lib.emap_image_fuse_arrays(em._ctx, KIND[kind], f32p(src), f32p(img), int(img.shape[0]), H, W, f32p(uv),
                                               va.ctypes.data_as(ct.POINTER(ct.c_uint8)), ct.c_double(float(alpha)), f32p(out)))
        np.asarray(new_sem_map).reshape(-1, C, C)[layer] = out
    kernel.__name__ = kind + "_correspondences_to_map_kernel"
    return kernel


def average_correspondences_to_map_kernel(width, height):
    return _sampler("average", width, height)


def exponential_correspondences_to_map_kernel(width, height, alpha):
    return _sampler("exponential", width, height, alpha)


def color_correspondences_to_map_kernel(width, height):
    return _sampler("color", width, height)
