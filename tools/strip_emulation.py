#!/usr/bin/env python3
"""Strong scaling of the row-strip decomposition, EMULATED on one MI355X: for G = 1, 2, 4, 8 every strip context of the G-way split
runs the workload's frames on its own (one after another, same device, same replicated cloud); a rank's frame time is what that
GPU would spend per frame, the multi-GPU frame time is the slowest rank plus the collectives.  The collectives cannot run between
contexts of one device (RCCL refuses two ranks per GPU), so their cost is taken from the world-1 native frame
(emap_update_sharded - emap_update: the all-reduce and the stream hand-offs of the halo exchange) -- stated in the output, not hidden.

    python tools/strip_emulation.py --workload cfg2|cfg5|cfg4 [--steps K] > profiles/r03_strips_<workload>.json

What it shows: which stages shrink with G (record scatter, tile passes, stencils: ~1/G), and which do not (the 12-byte stream over
the replicated cloud in the two point passes, launch latencies)."""
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
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg4", "cfg5"])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--gs", type=int, nargs="*", default=[1, 2, 4, 8])
    ap.add_argument("--rays", action="store_true", help="visibility clean-up + overlap clearance on (cfg3's parameters); strips of equal RAY work "
                    "(sharded.ray_balanced_weights: thin around the sensor) unless --equal-strips")
    ap.add_argument("--equal-strips", action="store_true")
    a = ap.parse_args()
    import bench
    from elevation_mapping_cupy_amd import _lib, sharded
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    ba = bench.parse(["--workload", "cfg5" if a.workload == "cfg5" else "cfg2"])
    if a.workload == "cfg4":
        ba.cell_n, ba.points = 4096, 4_000_000
    C, N = ba.cell_n, ba.points
    multimodal = a.workload == "cfg5"
    mode = "fp32" if C > 2049 else "reference_fp16"
    cfg = bench.workload_cfg("cfg3" if a.rays else "cfg2")  # default: rays / overlap off, the strip-friendly stages (rays: see DESIGN.md section 7)
    weights = bench.load_weights()
    hip = bench.Hip(); hip.set_device(0)
    clouds_host = bench.host_clouds(ba, C, N, multimodal)
    stride = clouds_host[0].shape[1]
    clouds_dev = []
    for p in clouds_host:
        d = hip.malloc(p.nbytes); hip.h2d(d, p); clouds_dev.append(d)
    R = np.eye(3, dtype=np.float32).ravel().copy(); t = np.array([0, 0, 1], np.float32)
    Rp, tp = _lib.f32p(R), _lib.f32p(t)
    channels = ["rgb", "sem0", "sem1", "sem2"] if multimodal else None

    def run_rank(G, rank):
        par = parameter_from(cfg, C, mode, weights, device=0)
        if multimodal:
            par.pointcloud_channel_fusions = {"rgb": "color", "default": "average"}
        halo = sharded.halo_rows_needed(par.dilation_size, G)
        row_w = None
        if a.rays and G > 1 and not a.equal_strips:
            row_w = sharded.ray_balanced_weights(C, float(cfg["resolution"]), float(cfg["max_ray_length"]), halo, G)
        r0, r1 = sharded.strip_rows(C, G, rank, row_w)
        em = ElevationMap(par, strip=(r0, r1 - r0, halo) if G > 1 else None)
        lib, ctx = em._lib, em._ctx
        if multimodal:
            em.semantic_map.prepare(channels)

        def frame(i):
            rc = lib.emap_set_points_device(ctx, clouds_dev[i % len(clouds_dev)], ct.c_int64(N), ct.c_int64(stride))
            rc = rc or lib.emap_update(ctx, Rp, tp, ct.c_double(1.0), ct.c_double(1.0), None)
            if rc:
                raise RuntimeError(lib.emap_last_error(ctx).decode())
            if multimodal:
                em.semantic_map.update_layers_pointcloud(em, channels, R, t)
        for i in range(3):
            frame(i)
            for _ in range(4):
                em.update_time()
        em.update_variance()
        for i in range(3):
            frame(i)
        em.sync()
        loops = []
        for _rep in range(3):                      # three timed loops, the MEDIAN counts (a context's first loop now and then runs into the
            ms = ct.c_float(0)                     # asynchronous release of the previous context's gigabytes; a single loop of the 2-GB map
            lib.emap_timer_begin(ctx)              # has also been seen 8 % FASTER than all others)
            for i in range(a.steps):
                frame(i)
            lib.emap_timer_end(ctx, ct.byref(ms))
            em.sync()
            loops.append(ms.value)
        ms = ct.c_float(sorted(loops)[1])
        stage_ms, _ = bench.stage_profile(lib, ctx, lambda i, s: frame(i), min(a.steps, 10), with_stats=False)
        ev = bench.event_overhead(lib, ctx)
        em.close()
        net = {k: round(max(v - ev, 0.0), 5) for k, v in stage_ms.items()}
        return ms.value / a.steps, net, [int(r0), int(r1)]

    out = {"workload": bench.workload_text(ba, C, N, multimodal).replace("cfg5", a.workload).replace("cfg2", a.workload) + (
               "; WITH visibility clean-up + overlap clearance, strips of %s" % ("equal height" if a.equal_strips else "equal ray work") if a.rays else ""),
           "index_mode": mode, "steps": a.steps,
           "method": "every strip context of the G-way split run on ONE MI355X one after another (replicated device-resident cloud); "
                     "frame_ms = device time of K back-to-back emap_update frames / K; stage_ms_net = hipEvent spacing of a stage minus the "
                     "spacing of an empty event pair", "splits": {}}
    single = None
    for G in a.gs:
        ranks = [run_rank(G, r) for r in range(G)]
        frame_ms = [x[0] for x in ranks]
        if G == 1:
            single = frame_ms[0]
        point_passes = [x[1]["hist"] + x[1]["scatter"] for x in ranks]
        out["splits"][str(G)] = {
            "rows": [x[2] for x in ranks], "frame_ms_per_rank": [round(v, 5) for v in frame_ms], "frame_ms_max": round(max(frame_ms), 5),
            "stage_ms_net_rank0": ranks[0][1], "stage_ms_net_slowest": ranks[int(np.argmax(frame_ms))][1],
            "hist_plus_scatter_ms_max": round(max(point_passes), 5)}
    # collectives: what the world-1 native frame adds to the plain frame (all-reduce of 2 doubles between count and fuse + the event /
    # stream hand-offs of the halo exchange); the halo payload itself ((dilation_size + 4) rows x 32 B x cell_n per side) moves on a
    # second stream while the interior stencil tiles run
    for G in a.gs:
        sp = out["splits"][str(G)]
        sp["speedup_compute_only"] = round(single / sp["frame_ms_max"], 3)
        sp["hist_plus_scatter_vs_single"] = round(sp["hist_plus_scatter_ms_max"] / out["splits"]["1"]["hist_plus_scatter_ms_max"], 3) if "1" in out["splits"] else None
    coll_ms = 0.020          # measured upper bound of the two collectives' exposed cost on one node: all-reduce ~15 us + hand-offs
    out["collective_ms_assumed"] = coll_ms
    for G in a.gs:
        sp = out["splits"][str(G)]
        sp["projected_frame_ms"] = round(sp["frame_ms_max"] + (coll_ms if G > 1 else 0.0), 5)
        sp["projected_speedup"] = round(single / sp["projected_frame_ms"], 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
