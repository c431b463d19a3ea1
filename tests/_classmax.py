"""Shared inputs of the ``pointcloud_class_max`` parity checks (tests/golden/make_golden.py: class_max_ref, tests/test_class_max_reference.py,
tests/test_hip_class_max.py): four frames on a 66 x 66 map whose class-id sets change from frame to frame -- the last one with sparse
ids, so that the reference's gather ``self.unique_id[elements_to_shift["id_max"]]`` (EM/fusion/pointcloud_class_max.py:85) indexes
beyond its table and CuPy's wrap-around decides what the id set becomes."""
import numpy as np

import _fixtures as fx

C, N = 66, 6000
CH = ["x", "y", "z", "top1", "top2"]
PCL_IDS, LAYER_IDS = [3, 4], [0, 1]
CLASS_SETS = [[0, 1, 2, 3], [1, 2, 3, 4, 5], [0, 2, 9], [7, 300, 2]]


def encode_max(prob, ids):
    """(half probability | class id << 16) bit-cast to float32: the wire format ClassMax.decode_max (:62-78) takes apart"""
    h = np.asarray(prob, np.float16).view(np.uint16).astype(np.uint32)
    return (h | (np.asarray(ids, np.uint32) << np.uint32(16))).view(np.float32)


def cloud(frame):
    """x y z | two (probability, id) channels.  The runner-up channel uses classes of its own: a class that is the maximum in SOME cell
    has its whole plane zeroed before the next layer (:119-121), so with shared classes the second layer would come out empty."""
    ids = CLASS_SETS[frame]
    rng = np.random.default_rng(600 + frame)
    p = fx.cloud(C, N, 610 + frame)
    k1 = rng.choice(ids, N); k2 = rng.choice(np.asarray(ids) + 100, N)
    pr1 = rng.uniform(0.5, 1.0, N); pr2 = rng.uniform(0.0, 0.04, N)
    return np.column_stack([p, encode_max(pr1, k1), encode_max(pr2, k2)]).astype(np.float32)


def clobbered(p, idx, valid, inside):
    """the cloud as add_points_kernel leaves it for the fusions (custom_kernels.py:260-262): columns 0..2 = cell index, valid, inside"""
    q = p.copy()
    q[:, 0] = idx.astype(np.float32); q[:, 1] = valid.astype(np.float32); q[:, 2] = inside.astype(np.float32)
    return q


def reference_frames(RefClassMax, point_index):
    """runs the reference's ClassMax (oracle/ref_fusion.py) over the frames; point_index(p) -> (idx, valid, inside).
    Returns per frame: the two semantic layers, the two id planes, the fusion's unique_id table."""
    cm = RefClassMax(C)
    sem = np.zeros((2, C, C), np.float32); new_map = np.zeros((2, C, C), np.float32)
    ets = {"id_max": np.zeros((2, C, C), np.uint32)}
    out = []
    for f in range(len(CLASS_SETS)):
        p = cloud(f)
        q = clobbered(p, *point_index(p))
        cm.fuse(q, None, None, np.array(PCL_IDS, np.int32), np.array(LAYER_IDS, np.int32), None, sem, new_map, ets)
        out.append((sem.copy(), ets["id_max"].copy(), np.asarray(cm.unique_id).astype(np.uint32).copy()))
    return out
