#!/usr/bin/env python3
"""The visibility pass away from its steady state: (a) the frames after clear() -- every visit of an unknown cell lowers an upper bound;
(b) the frame after a long occlusion (--ticks update_time() calls without a cloud: every cell stale) -- every visit that passes the
penetration test decrements a cell.  Prints one JSON line: k_rays per frame (event spacing) for the steady state, the frames after
clear() and the frame after the occlusion.  A/B of library builds: EMAP_HIP_LIB=tools/ab/x.so python tools/exp_ray_storm.py"""
import argparse
import ctypes as ct
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cell-n", type=int, default=1024)
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--ticks", type=int, default=12)
    a = ap.parse_args()
    import bench
    import _fixtures as fx
    from elevation_mapping_cupy_amd import _lib
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    C, N = a.cell_n, a.points
    cfg = bench.workload_cfg("cfg3")
    em = ElevationMap(parameter_from(cfg, C, "fp32" if C > 2049 else "reference_fp16", bench.load_weights()))
    lib, ctx = em._lib, em._ctx
    hip = bench.Hip(); hip.set_device(0)
    clouds = []
    for s in range(4):
        p = fx.cloud(C, N, s, dz=-0.02 * s)
        d = hip.malloc(p.nbytes); hip.h2d(d, p); clouds.append(d)
    R = np.eye(3, dtype=np.float32).ravel().copy(); t = np.array([0, 0, 1], np.float32)
    Rp, tp = _lib.f32p(R), _lib.f32p(t)

    def frame(i):
        rc = lib.emap_set_points_device(ctx, clouds[i % 4], ct.c_int64(N), ct.c_int64(3))
        rc = rc or lib.emap_update(ctx, Rp, tp, ct.c_double(1.0), ct.c_double(1.0), None)
        if rc:
            raise RuntimeError(lib.emap_last_error(ctx).decode())

    def rays_ms(i):
        frame(i)
        ms10 = (ct.c_float * 10)()
        lib.emap_get_stage_times(ctx, ms10)
        return float(ms10[6])
    for i in range(8):
        frame(i); em.update_time()
    em.sync()
    lib.emap_enable_stage_timing(ctx, 1)
    steady = []
    for i in range(10):
        steady.append(rays_ms(i)); em.update_time()
    storm = []
    for rep in range(3):
        for _ in range(a.ticks):
            em.update_time()
        storm.append(rays_ms(rep))
        for i in range(4):
            rays_ms(i); em.update_time()
    em.clear()
    cold = [rays_ms(i) for i in range(6)]
    med = float(np.median(steady))
    print(json.dumps({"lib": os.environ.get("EMAP_HIP_LIB", "in-tree"), "cell_n": C, "points": N, "steady_rays_ms": round(med, 4),
                      "after_occlusion_ms": [round(x, 4) for x in storm], "after_clear_ms": [round(x, 4) for x in cold],
                      "occlusion_over_steady": round(max(storm) / med, 2), "clear_over_steady": round(max(cold) / med, 2)}), flush=True)
    em.close()


if __name__ == "__main__":
    main()
