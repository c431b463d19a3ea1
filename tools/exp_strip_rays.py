"""GPU experiment: device time of the visibility pass on ONE strip of a W-way split of the 1024^2 / 1 M workload (no exchange
needed for timing).  Shows what the per-strip clipping of the step range buys: without it every rank marches all 353 M steps."""
import ctypes as ct
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _fixtures as fx  # noqa: E402
from elevation_mapping_cupy_amd.configs import CORE_PARAM_YAML, parameter_from  # noqa: E402
from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap  # noqa: E402
from elevation_mapping_cupy_amd.sharded import halo_rows_needed, ray_balanced_weights, strip_rows  # noqa: E402

C, N = 1024, 1_000_000
w = np.load(os.path.join(ROOT, "tests", "golden", "weights.npz"))
weights = {k: w[k] for k in ("w1", "w2", "w3", "w_out")}
R = np.eye(3, dtype=np.float32); t = np.array([0, 0, 1], np.float32)
clouds = [fx.cloud(C, N, s, dz=(0.0 if s == 0 else -0.02 * s)) for s in range(3)]
CASES = [(1, 0, False), (8, 4, False), (8, 3, False), (8, 2, False), (8, 0, False), (8, 4, True), (8, 3, True), (8, 1, True), (8, 0, True)]
for W, rank, balanced in CASES:
    H_ = halo_rows_needed(CORE_PARAM_YAML["dilation_size"], W)
    row_w = ray_balanced_weights(C, 0.04, CORE_PARAM_YAML["max_ray_length"], H_, W) if balanced else None
    r0, r1 = strip_rows(C, W, rank, row_w)
    H = halo_rows_needed(CORE_PARAM_YAML["dilation_size"], W)
    m = ElevationMap(parameter_from(dict(CORE_PARAM_YAML), C, "reference_fp16", weights), strip=(r0, r1 - r0, H) if W > 1 else None)
    ms = ct.c_float(0)
    acc = []
    for f in range(16):
        m.bind_points(clouds[f % 3])
        m.stage("count", R, t)
        m._chk(m._lib.emap_set_drift_inputs(m._ctx, ct.c_double(1.0), ct.c_double(1.0), None, None))
        m.stage("fuse", R, t); m.stage("commit")
        m._chk(m._lib.emap_timer_begin(m._ctx))
        m.stage("rays", R, t)
        m._chk(m._lib.emap_timer_end(m._ctx, ct.byref(ms)))
        m.stage("average")
        if f < 3:                      # warm-up exactly like bench.py: time ticks only between the first frames
            for _ in range(4):
                m.update_time()
        if f >= 6:
            acc.append(ms.value)
    print("strips %d %s rank %d (rows %d..%d): rays %.3f ms" % (W, "balanced" if balanced else "uniform", rank, r0, r1, float(np.mean(acc))))
