"""``pointcloud_class_max`` (reference EM/fusion/pointcloud_class_max.py:50-126): the top-n classes of a point arrive as n channels,
each a float whose low 16 bits are the class probability (IEEE half) and whose high 16 bits are the class id (``decode_max``,
:62-78); per frame the layers of the fusion get, per cell, the n largest summed probabilities (normalised) and the map keeps the
class id of each in ``elements_to_shift["id_max"]``.  The whole frame runs on the device behind ``emap_semantic_class_max``
(csrc/emap_semantic.hip: exact integer sums, per-layer maximum, the reference's whole-plane zeroing between layers); the object
keeps what the reference's object keeps across frames: ``unique_id``, the sorted class ids seen so far in the map and the cloud."""
from __future__ import annotations

import ctypes as ct

import numpy as np

from .._lib import f32p
from .fusion_manager import FusionBase


def encode_max(prob, class_id):
    """the sensor side of ``decode_max``: (probability, id) -> the float the cloud carries"""
    bits = np.asarray(prob, np.float16).view(np.uint16).astype(np.uint32) | (np.asarray(class_id, np.uint32) << np.uint32(16))
    return bits.view(np.float32)


class ClassMax(FusionBase):
    def __init__(self, params, *args, **kwargs):
        self.name = "pointcloud_class_max"
        self.kind = "class_max"
        self.cell_n = params.cell_n
        self.resolution = params.resolution
        self.unique_id = np.array([0], np.uint32)          # :59

    def fuse(self, emap, pcl_ids, layer_ids, R, t):
        """one frame on the cloud bound to ``emap`` (the reference's ``__call__``, :80-126)"""
        n = len(pcl_ids)
        if n == 0:
            return
        if n > 8:
            raise ValueError("class_max: at most 8 (probability, id) channels per cloud")
        ch = np.ascontiguousarray(pcl_ids, np.int32); ly = np.ascontiguousarray(layer_ids, np.int32)
        prev = np.ascontiguousarray(self.unique_id, np.uint32)
        out = np.empty(65536, np.uint32); cnt = ct.c_int32(0)
        i32p, u32p = ct.POINTER(ct.c_int32), ct.POINTER(ct.c_uint32)
        R = np.ascontiguousarray(np.asarray(R, np.float32).reshape(9)); t = np.ascontiguousarray(np.asarray(t, np.float32).reshape(3))
        emap._chk(emap._lib.emap_semantic_class_max(emap._ctx, f32p(R), f32p(t), n, ch.ctypes.data_as(i32p), ly.ctypes.data_as(i32p),
                                                    prev.ctypes.data_as(u32p), int(prev.size), out.ctypes.data_as(u32p), int(out.size),
                                                    ct.byref(cnt)))
        self.unique_id = out[:cnt.value].copy()
