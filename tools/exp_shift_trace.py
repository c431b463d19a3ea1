"""GPU box, under `rocprofv3 --kernel-trace --stats`: what a map shift costs.  Three frames on a 1024^2 map, `move_to` by 3 rows /
5 columns between them (and once with a semantic layer): the kernel list shows that the shift itself launches nothing on the core
map (circular origin + pending-move replay) and one band-clear kernel per shift for semantic layers."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _fixtures as fx  # noqa: E402
from _util import make_parameter  # noqa: E402
from oracle import emap_oracle as eo  # noqa: E402  (parameter sets only)
from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap  # noqa: E402

C = 1024
w = np.load(os.path.join(ROOT, "tests", "golden", "weights.npz"))
w = {k: w[k] for k in ("w1", "w2", "w3", "w_out")}
em = ElevationMap(make_parameter(dict(eo.YAML, enable_visibility_cleanup=False), C, "reference_fp16", w))
R, t = fx.POSES["identity"]
res = em.resolution
MOVE = os.environ.get('EXP_MOVE', '1') == '1'
for f in range(6):
    em.update_map_with_kernel(fx.cloud(C, 1_000_000, f), [], R, t.copy(), 0.0, 0.0)
    em.sync()
    if MOVE:
        em.move_to(np.array([3 * res * (f + 1), 5 * res * (f + 1), 0.0]), np.eye(3))
        em.sync()
print("org", em.logical_row_begin, float(em.elevation_map[2].sum()))
