#!/usr/bin/env python3
"""Does the stencil kernel (and the tile kernels that walk the same strides) degrade with the map's ROW PITCH?  Round 3 measured the
LDS-DMA variant of k_post 12 % ahead at 4096^2, 21 % behind at 6144^2 and 60 % behind at 8192^2 and could not say why; a row of a
power-of-two map is a power-of-two number of bytes (8192 x 16 B = 128 KB), the classic recipe for channel / bank aliasing between the
rows of a tile.  This tool times the stage on maps whose side is NOT such a number next to the ones that are:

    for C in 4096 4160 6144 6208 8192 8256; do for V in post dma; do
      [ $V = dma ] && export EMAP_POST_DMA_WINDOW="1 100000" || export EMAP_POST_DMA=0
      python tools/exp_post_pitch.py --cell-n $C --tag $V; unset EMAP_POST_DMA_WINDOW EMAP_POST_DMA; done; done

(one process per variant: the selection knobs are read once).  Prints one JSON line: ns per cell of the stencil launch and the
event-spaced stages of a whole frame."""
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
    ap.add_argument("--cell-n", type=int, required=True)
    ap.add_argument("--tag", default="")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--density", type=float, default=0.24, help="points per cell and frame")
    a = ap.parse_args()
    import bench
    import _fixtures as fx
    from elevation_mapping_cupy_amd import _lib
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    C = a.cell_n
    N = int(C * C * a.density)
    cfg = bench.workload_cfg("cfg2")
    em = ElevationMap(parameter_from(cfg, C, "fp32" if C > 2049 else "reference_fp16", bench.load_weights()))
    lib, ctx = em._lib, em._ctx
    hip = bench.Hip(); hip.set_device(0)
    clouds = []
    for s in range(2):
        p = fx.cloud(C, N, s, dz=-0.02 * s)
        d = hip.malloc(p.nbytes); hip.h2d(d, p); clouds.append(d)
    R = np.eye(3, dtype=np.float32).ravel().copy(); t = np.array([0, 0, 1], np.float32)
    Rp, tp = _lib.f32p(R), _lib.f32p(t)

    def frame(i, stats=None):
        rc = lib.emap_set_points_device(ctx, clouds[i % 2], ct.c_int64(N), ct.c_int64(3))
        rc = rc or lib.emap_update(ctx, Rp, tp, ct.c_double(1.0), ct.c_double(1.0), stats)
        if rc:
            raise RuntimeError(lib.emap_last_error(ctx).decode())
    for i in range(4):
        frame(i)
    em.sync()
    ev = bench.event_overhead(lib, ctx)
    loops = []
    for _ in range(3):
        ms = ct.c_float(0)
        lib.emap_timer_begin(ctx)
        for _k in range(a.reps):
            lib.emap_post(ctx)
        lib.emap_timer_end(ctx, ct.byref(ms))
        loops.append(ms.value / a.reps)
    post_ms = float(np.median(loops))
    st, _ = bench.stage_profile(lib, ctx, frame, 6, with_stats=False)
    L = C * C
    print(json.dumps({"cell_n": C, "tag": a.tag, "row_pitch_bytes_halfcells": 16 * C, "points": N, "post_ms": round(post_ms, 5),
                      "post_ns_per_cell": round(post_ms * 1e6 / L, 4), "post_algorithmic_TBs": round(40 * L / (post_ms * 1e-3) / 1e12, 3),
                      "stage_ns_per_cell": {k: round(max(v - ev, 0) * 1e6 / L, 4) for k, v in st.items() if v > 0}}), flush=True)
    em.close()


if __name__ == "__main__":
    main()
