#!/usr/bin/env python3
"""Frames of the cfg3 workload (visibility pass + overlap clearance, 1024^2 map) on the ray-cast terrain scene ONLY -- for a profiler:
    rocprofv3 --kernel-trace --stats -- python tools/exp_terrain_frame.py --frames 30
prints one JSON line (ms per frame, stage spacings, valid cells); --scene noise runs the white-noise clouds of the benchmark instead."""
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
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--scene", default="terrain", choices=["terrain", "noise"])
    ap.add_argument("--cell-n", type=int, default=1024)
    a = ap.parse_args()
    import bench
    import _fixtures as fx
    from elevation_mapping_cupy_amd import _lib
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    C = a.cell_n
    em = ElevationMap(parameter_from(bench.workload_cfg("cfg3"), C, "fp32" if C > 2049 else "reference_fp16", bench.load_weights()))
    lib, ctx = em._lib, em._ctx
    hip = bench.Hip(); hip.set_device(0)
    if a.scene == "terrain":
        host = [fx.terrain_cloud(C, 2000, 500, s, shift=sh) for s, sh in enumerate((0.0, 0.4, -0.3, 0.2))]
    else:
        host = [fx.cloud(C, 1_000_000, s, dz=-0.02 * s) for s in range(4)]
    N = host[0].shape[0]
    dev = []
    for p in host:
        d = hip.malloc(p.nbytes); hip.h2d(d, p); dev.append(d)
    R = np.eye(3, dtype=np.float32).ravel().copy(); t = np.array([0, 0, 1], np.float32)
    Rp, tp = _lib.f32p(R), _lib.f32p(t)

    def frame(i, stats=None):
        rc = lib.emap_set_points_device(ctx, dev[i % len(dev)], ct.c_int64(N), ct.c_int64(3))
        rc = rc or lib.emap_update(ctx, Rp, tp, ct.c_double(1.0), ct.c_double(1.0), stats)
        if rc:
            raise RuntimeError(lib.emap_last_error(ctx).decode())
    for i in range(3):
        frame(i)
        for _ in range(4):
            em.update_time()
    for i in range(5):
        frame(i)
    em.sync()
    ms = ct.c_float(0)
    lib.emap_timer_begin(ctx)
    for i in range(a.frames):
        frame(i)
    lib.emap_timer_end(ctx, ct.byref(ms))
    st, visits = bench.stage_profile(lib, ctx, frame, 6)
    print(json.dumps({"scene": a.scene, "points": N, "ms_per_frame": round(ms.value / a.frames, 5), "ray_visits": int(visits),
                      "valid_fraction": round(float((em.get_layer_raw(2) > 0.5).mean()), 4),
                      "stage_us": {k: round(v * 1e3, 1) for k, v in st.items() if v > 0}}), flush=True)
    em.close()


if __name__ == "__main__":
    main()
