"""GPU box: k_post time (event spacing, us) per frame while the map is being moved.  EXP_DR / EXP_DC = rows / columns per move."""
import ctypes as ct
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _fixtures as fx  # noqa: E402
from _util import make_parameter  # noqa: E402
from oracle import emap_oracle as eo  # noqa: E402
from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap  # noqa: E402

C = 1024
w = np.load(os.path.join(ROOT, "tests", "golden", "weights.npz"))
w = {k: w[k] for k in ("w1", "w2", "w3", "w_out")}
for dr, dc, warm in ((0, 0, 0), (3, 0, 0), (0, 5, 0), (3, 5, 0), (3, 5, 12), (16, 64, 0), (1, 1, 0)):
    em = ElevationMap(make_parameter(dict(eo.YAML, enable_visibility_cleanup=False), C, "reference_fp16", w))
    R, t = fx.POSES["identity"]
    for f in range(warm):
        em.update_map_with_kernel(fx.cloud(C, 1_000_000, f % 5), [], R, t.copy(), 0.0, 0.0)
    em._lib.emap_enable_stage_timing(em._ctx, 1)
    out = []
    for f in range(6):
        em.update_map_with_kernel(fx.cloud(C, 1_000_000, f % 5), [], R, t.copy(), 0.0, 0.0)
        ms = (ct.c_float * 10)()
        em._lib.emap_get_stage_times(em._ctx, ms)
        out.append(round(ms[9] * 1e3, 1))
        if dr or dc:
            em.shift_map_xy(np.array([dr, dc]))
    print("shift per frame (%d rows, %d cols), %d warm frames: post us =" % (dr, dc, warm), out, "holes left:", int((em.elevation_map[2] < 0.5).sum()))
    em.close()
