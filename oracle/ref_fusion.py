"""TEST INFRASTRUCTURE ONLY.  Executes the reference's HOST code of the ``pointcloud_class_max`` fusion -- ``ClassMax.decode_max`` and
``ClassMax.__call__`` (EM/fusion/pointcloud_class_max.py:62-126) -- from the file where it lies under /root/reference, the way
``oracle/ref_host.py`` does for ``elevation_mapping.py``: the two function bodies are taken from the syntax tree unmodified, vetted by
the same allow-list (``ref_host._vet``: no imports, no nested definitions, calls rooted at self / cp / np / locals / a few builtins)
and compiled with a NumPy-backed stand-in for ``cp``; the fusion's own ``sum_max_kernel`` (:12-47 of the same file) is the REFERENCE'S
kernel source compiled for the host (``oracle/build_ref.py``, parameter sets with ``class_max_kernel``).  Nothing of the reference
is copied into the repository; live execution is opt-in (``EMAP_REF_EXEC=1``), the committed golden vectors
(tests/golden/class_max_ref66.npz, made by tests/golden/make_golden.py) are the default pin.

Where CuPy and NumPy differ the stand-in follows CuPy, because that is what the reference runs on:
* integer-array indexing with out-of-bounds positions WRAPS AROUND in CuPy (documented difference; NumPy raises): the gather
  ``self.unique_id[elements_to_shift["id_max"]]`` (:85) indexes the id table with stored class VALUES, so it relies on this;
* ``cp.bitwise_and(x, 0xFFFF, dtype=np.uint16)`` (:74) converts to the requested dtype (NumPy refuses the uint32 -> uint16 cast)."""
from __future__ import annotations

import ast
import os
import types

import numpy as np

from . import build_ref, ref_host

REF_FILE = os.environ.get("EMAP_REF_FUSION_FILE", os.path.join(build_ref.REF_ROOT, "fusion", "pointcloud_class_max.py"))


def available():
    return os.environ.get("EMAP_REF_EXEC", "0") == "1" and os.path.isfile(REF_FILE)


class CpArray(np.ndarray):
    """ndarray whose integer-array gathers wrap out-of-bounds positions, like CuPy's"""

    def __getitem__(self, key):
        if isinstance(key, np.ndarray) and key.dtype.kind in "iu" and self.ndim == 1:
            return np.take(np.asarray(self), key, mode="wrap").view(CpArray)
        return super().__getitem__(key)


def _cp_namespace():
    xp = types.ModuleType("numpy_as_cupy")
    xp.__dict__.update(np.__dict__)
    xp.unique = lambda a: np.unique(np.asarray(a)).view(CpArray)
    xp.array = lambda *a, **k: np.array(*a, **k).view(CpArray)
    xp.bitwise_and = lambda a, b, dtype=None: (np.bitwise_and(np.asarray(a), b).astype(dtype) if dtype is not None else np.bitwise_and(a, b))
    return {"cp": xp, "np": np, "__builtins__": ref_host.exec_builtins(enumerate=enumerate)}


def load(ref_kernels):
    """returns the class RefClassMax(cell_n); ``ref_kernels``: an oracle.ref_kernels.RefKernels of a parameter set with
    ``class_max_kernel`` (the fusion file's own sum_max_kernel compiled for the host)"""
    tree = ast.parse(open(REF_FILE).read(), REF_FILE)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "ClassMax")
    fns = {n.name: n for n in cls.body if isinstance(n, ast.FunctionDef)}
    body = []
    for name, new in (("decode_max", "decode_max"), ("__call__", "fuse")):
        fn = fns[name]
        ref_host._vet(fn, "ClassMax." + name, extra_builtins=("enumerate",))
        fn.name = new
        body.append(fn)
    ns = _cp_namespace()
    exec(compile(ast.fix_missing_locations(ast.Module(body=body, type_ignores=[])), REF_FILE, "exec"), ns)

    def kernel(points_all, max_pt, max_id, pcl_ids, layer_ids, pcl_channels, prob_sum, size):
        """the call of :94-103 on the compiled reference kernel (sequential element order: one legal order of its float atomics)"""
        p = np.ascontiguousarray(points_all, np.float32)
        mp, mi = np.ascontiguousarray(max_pt, np.float32), np.ascontiguousarray(max_id, np.uint32)
        pc, ly, ch = (np.ascontiguousarray(a, np.int32) for a in (pcl_ids, layer_ids, pcl_channels))
        assert prob_sum.dtype == np.float32 and prob_sum.flags["C_CONTIGUOUS"]
        ref_kernels._call("cmax_sum_max", [p, mp, mi, pc, ly, ch, prob_sum], size)

    class RefClassMax:
        """state of the reference's ClassMax.__init__ (:50-60)"""

        def __init__(self, cell_n):
            self.name = "pointcloud_class_max"
            self.cell_n = int(cell_n)
            self.sum_max_kernel = kernel
            self.unique_id = ns["cp"].array([0])

    RefClassMax.decode_max = ns["decode_max"]
    RefClassMax.fuse = ns["fuse"]
    return RefClassMax
