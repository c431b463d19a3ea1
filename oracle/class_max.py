"""TEST INFRASTRUCTURE (oracle): NumPy restatement of the reference fusion ``pointcloud_class_max``
(EM/fusion/pointcloud_class_max.py:50-126), statement by statement -- ``cp`` -> ``np``; the elementwise ``sum_max_kernel`` (:12-47)
is the loop in ``sum_max`` (pinned against the reference's own kernel source by tests/test_oracle_golden.py /
tests/test_hip_semantic_factories.py through ``oracle/ref_kernels.py: sem_sum_max``).  Only tests may import this file.

Two deliberate definitions where the reference is not defined:
* the probability sums are the EXACT sums rounded once to float32 (the reference's float atomics round after every addition, in an
  order the GPU picks); every half is a multiple of 2^-24, so float64 accumulation is exact here;
* ``self.unique_id[elements_to_shift["id_max"]]`` (:85) gathers with class VALUES as positions; positions beyond the table WRAP
  AROUND, as CuPy's integer-array indexing does (a documented difference from NumPy, which raises) -- since round 4; before, they
  were ignored.

Round 4: the reference's own statements are executed from its file as well (oracle/ref_fusion.py -> tests/golden/class_max_ref66.npz);
tests/test_class_max_reference.py compares this restatement with them: ids and the id table exact, probabilities within float32
summation order (the one deviation that remains, by design: exact sums)."""
import numpy as np


def decode_max(mer):
    """:62-78"""
    mer = np.ascontiguousarray(mer, np.float32).view(np.uint32)
    ma = (mer & np.uint32(0xFFFF)).astype(np.uint16).view(np.float16).astype(np.float32)
    ind = mer >> np.uint32(16)
    return ma, ind


class ClassMaxOracle:
    def __init__(self, cell_n):
        self.cell_n = cell_n
        self.unique_id = np.array([0], np.uint32)                     # :59

    def __call__(self, points_all, idx, valid, inside, pcl_ids, layer_ids, semantic_map, id_max):
        """``points_all`` (N, 3 + K) raw cloud, (idx, valid, inside) what the tail of add_points_kernel leaves in its first three
        columns (custom_kernels.py:260-262); ``semantic_map`` (L, C, C) float32 and ``id_max`` (len(layer_ids), C, C) uint32 are
        updated in place (id_max[i] belongs to layer_ids[i])."""
        C = self.cell_n
        max_pt, pt_id = decode_max(points_all[:, pcl_ids])            # :81
        unique_idm = np.unique(pt_id)                                  # :83
        stored = id_max.reshape(-1).astype(np.int64) % self.unique_id.size      # positions beyond the table wrap around (CuPy)
        unique_ida = np.unique(self.unique_id[stored])                 # :85
        self.unique_id = np.unique(np.concatenate((unique_idm, unique_ida))).astype(np.uint32)      # :86
        prob = np.zeros((len(self.unique_id), C * C), np.float64)     # :88 (float64: exact)
        pos = np.searchsorted(self.unique_id, pt_id)                   # :90-92
        ok = (valid != 0) & (inside != 0)                              # sum_max_kernel :33-34
        cells = idx[ok].astype(np.int64)
        for it in range(pt_id.shape[1]):
            p = max_pt[ok, it].astype(np.float64)
            fin = np.isfinite(p)
            np.add.at(prob, (pos[ok, it][fin], cells[fin]), p[fin])
        prob_sum = prob.astype(np.float32).reshape(len(self.unique_id), C, C)
        new_map = np.zeros((len(layer_ids), C, C), np.float32)
        for i, lay in enumerate(layer_ids):                            # :113-116
            am = np.argmax(prob_sum, axis=0)
            new_map[i] = np.max(prob_sum, axis=0)
            id_max[i] = self.unique_id[am]
            prob_sum[np.unique(am)] = 0                                # whole planes: every class that is a maximum somewhere
        sum_alpha = np.zeros((C, C), np.float32)
        for i in range(len(layer_ids)):                                # :118 (sequential float adds)
            sum_alpha = sum_alpha + new_map[i]
        sum_alpha[sum_alpha == 0] = 1                                  # :120
        for i, lay in enumerate(layer_ids):
            semantic_map[lay] = new_map[i] / sum_alpha                 # :121
