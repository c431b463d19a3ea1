#!/usr/bin/env python3
"""Strong scaling of the row-strip decomposition, EMULATED on one MI355X -- the frame every number below times is the library's own
sharded frame, emap_update_sharded: count -> all-reduce on the strip's stream -> gate folded into the tile kernel -> fuse [-> rays] ->
halo exchange on the second stream next to the interior stencil tiles -> event wait -> boundary tiles (+ the per-strip RGB / semantic
fusion of multi-modal clouds).  RCCL refuses two ranks on one device, so the ten RCCL entry points are served by the stream-ordered
in-process stand-in tests/fake_rccl/stream_rccl.hip (events across streams, a reduction kernel, device-to-device copies: the same
stream structure as under RCCL, no host synchronisation).  Three measurements per split G:

  solo      every rank of the G-way split runs ALONE on the GPU, one after another (stand-in in loop-back mode: the rank's collectives
            move the same bytes through the same streams and events, nobody else is there): frame_ms_per_rank = what that GPU would
            spend per frame on everything but the wire; the multi-GPU frame is bounded below by the slowest rank.
  lockstep  all G ranks ALIVE at once (threads), their collectives really meet: the all-reduce of a frame completes when the SLOWEST
            rank has delivered its sums, the halo rows really come from the neighbours.  The ranks share the one GPU, so the wall time
            of a frame is the time of ALL ranks' work: lockstep_ms / single_ms = the work inflation of the decomposition (replicated
            cloud streams, halo re-staging, launches), G x single_ms / lockstep_ms an upper bound of the speed-up that does not depend
            on any per-rank timing.
  wire      what one GPU cannot show: the latency of an 8-rank all-reduce of 16 bytes and of the halo rows over xGMI.  The real RCCL
            is measured at world size 1 (its launch + kernel cost, a LOWER bound of the multi-rank latency) and the projection is given
            for several assumed wire latencies instead of one.

    python tools/strip_emulation.py --workload cfg2|cfg4|cfg5 [--rays] [--steps K] > profiles/r04_strips_<workload>.json"""
import argparse
import ctypes as ct
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg4", "cfg5"])
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--gs", type=int, nargs="*", default=[1, 2, 4, 8])
    ap.add_argument("--rays", action="store_true", help="visibility clean-up + overlap clearance on (cfg3's parameters); strips of equal RAY work "
                    "(sharded.ray_balanced_weights: thin around the sensor) unless --equal-strips")
    ap.add_argument("--equal-strips", action="store_true")
    ap.add_argument("--no-lockstep", action="store_true")
    ap.add_argument("--skip-single", action="store_true", help="profiling aid: run the single context only briefly (its kernels would mix into a rocprofv3 --stats summary of the strips)")
    ap.add_argument("--interleaved-cloud", action="store_true", help="cfg5: interleaved (N, 7) device rows instead of the de-interleaved layout of an uploaded cloud")
    ap.add_argument("--replicated-cloud", action="store_true", help="every rank binds the WHOLE cloud (rounds 1-4); default: a rank binds only the points "
                    "that can land in its rows wherever the frame allows it (emap_strip_point_mask = what emap_upload_points_strip uploads)")
    ap.add_argument("--scene", default="uniform", choices=["uniform", "terrain"], help="terrain (1024^2 only): the scan-ordered, ray-cast scene of "
                    "tests/_fixtures.py: terrain_cloud -- one heavy tile under the sensor, 86 %% of the cells never seen")
    ap.add_argument("--ray-mode", default="auto", choices=["auto", "by_row", "by_ray"], help="with --rays: how the sharded frame runs the visibility "
                    "pass (emap_set_ray_mode; auto = by ray from 2048^2 cells on)")
    a = ap.parse_args()
    import bench
    from _util import rccl_stand_in
    from elevation_mapping_cupy_amd import _lib, sharded
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    ba = bench.parse(["--workload", "cfg5" if a.workload == "cfg5" else "cfg2"])
    if a.workload == "cfg4":
        ba.cell_n, ba.points = 4096, 4_000_000
    C, N = ba.cell_n, ba.points
    multimodal = a.workload == "cfg5"
    mode = "fp32" if C > 2049 else "reference_fp16"
    cfg = bench.workload_cfg("cfg3" if a.rays else "cfg2")
    weights = bench.load_weights()
    hip = bench.Hip(); hip.set_device(0)
    if a.scene == "terrain":
        import _fixtures as fx
        assert C == 1024 and not multimodal, "the terrain scene is built for the 1024^2 map"
        clouds_host = [fx.terrain_cloud(C, 2000, 500, s_, shift=sh) for s_, sh in enumerate((0.0, 0.4, -0.3, 0.2))]
        N = clouds_host[0].shape[0]
    else:
        clouds_host = bench.host_clouds(ba, C, N, multimodal)
    clouds_dev = bench.device_clouds(hip, clouds_host, not a.interleaved_cloud)
    R = np.eye(3, dtype=np.float32).ravel().copy(); t = np.array([0, 0, 1], np.float32)
    Rp, tp = _lib.f32p(R), _lib.f32p(t)
    channels = ["rgb", "sem0", "sem1", "sem2"] if multimodal else None
    stand_in = rccl_stand_in("stream").encode()
    by_ray = a.rays and (a.ray_mode == "by_ray" or (a.ray_mode == "auto" and C >= 2048))

    def make_rank(G, rank, loopback):
        """strip context + communicator of one rank; returns (map, frame function)"""
        par = parameter_from(cfg, C, mode, weights, device=0)
        if multimodal:
            par.pointcloud_channel_fusions = {"rgb": "color", "default": "average"}
        halo = sharded.halo_rows_needed(par.dilation_size, G)
        row_w = None
        if a.rays and G > 1 and not a.equal_strips and not by_ray:      # (by ray: every rank marches the rays of its own points -- equal heights)
            row_w = sharded.ray_balanced_weights(C, float(cfg["resolution"]), float(cfg["max_ray_length"]), halo, G)
        r0, r1 = sharded.strip_rows(C, G, rank, row_w)
        em = ElevationMap(par, strip=(r0, r1 - r0, halo) if G > 1 else None)
        lib, ctx = em._lib, em._ctx
        if multimodal:
            em.semantic_map.prepare(channels)
        em._rows01 = [int(r0), int(r1)]
        em.set_ray_mode(a.ray_mode)
        # the rank's clouds: the whole ones, or -- where the frame allows it -- only the points that can land in its rows
        em._clouds, em._n_local, em._bucketed = clouds_dev, [N] * len(clouds_dev), False
        if G > 1 and not a.replicated_cloud and (not a.rays or by_ray):
            local = []
            for p_ in clouds_host:
                q_ = np.ascontiguousarray(p_[em.strip_point_mask(p_, R, t)])
                local.append(q_ if q_.shape[0] else np.full((1, p_.shape[1]), np.nan, np.float32))
            em._n_local = [q_.shape[0] for q_ in local]
            em._clouds = bench.device_clouds(hip, local, not a.interleaved_cloud)
            em._bucketed = True
        return em

    def drop_rank(em):
        if em._bucketed:
            bench.free_clouds(hip, em._clouds)
        em.close()

    def frame_fn(em, sharded_frame):
        lib, ctx = em._lib, em._ctx
        call = lib.emap_update_sharded if sharded_frame else lib.emap_update

        def frame(i, stats=None):
            k = i % len(em._clouds)
            rc = bench.bind_cloud(lib, ctx, em._clouds[k], em._n_local[k])
            if em._bucketed:
                rc = rc or lib.emap_declare_points_bucketed(ctx, Rp, tp, ct.c_int64(N))
            in_frame = multimodal and bench.sem_in_frame(lib)
            if in_frame and not rc:          # the RGB / semantic fusion rides inside the frame's tile pass (emap_frame_semantics)
                em.semantic_map.declare_frame(em, channels)
            rc = rc or call(ctx, Rp, tp, ct.c_double(1.0), ct.c_double(1.0), stats)
            if rc:
                raise RuntimeError(lib.emap_last_error(ctx).decode())
            if multimodal and not in_frame:
                em.semantic_map.update_layers_pointcloud(em, channels, R, t)
        return frame

    def warm(em, frame):
        for i in range(3):
            frame(i)
            for _ in range(4):
                em.update_time()
        em.update_variance()
        for i in range(3):
            frame(i)
        em.sync()

    def timed_loops(em, frame, steps, reps=3):
        lib, ctx = em._lib, em._ctx
        loops = []
        for _rep in range(reps):                   # the MEDIAN of the timed loops counts (a context's first loop now and then runs into the
            ms = ct.c_float(0)                     # asynchronous release of the previous context's gigabytes)
            lib.emap_timer_begin(ctx)
            for i in range(steps):
                frame(i)
            lib.emap_timer_end(ctx, ct.byref(ms))
            em.sync()
            loops.append(ms.value / steps)
        return float(np.median(loops))

    def comm_init(em, uid, rank, G):
        if em._lib.emap_comm_init(em._ctx, stand_in, uid, rank, G):
            raise RuntimeError(em._lib.emap_last_error(em._ctx).decode())

    def new_uid(lib):
        uid = (ct.c_uint8 * 128)()
        assert lib.emap_comm_unique_id(stand_in, uid) == 0
        return uid

    # ---- single context: the plain frame (emap_update) and the world-1 sharded frame over the REAL RCCL ------------------------------
    em = make_rank(1, 0, False)
    f_plain = frame_fn(em, False)
    if a.skip_single:                 # (profiling runs of the strips alone: rocprofv3 --stats aggregates by kernel name)
        warm = lambda e, f: [f(i) for i in range(2)] and e.sync()      # noqa: E731
    warm(em, f_plain)
    single_ms = timed_loops(em, f_plain, a.steps if not a.skip_single else 1, reps=3 if not a.skip_single else 1)
    ev = bench.event_overhead(em._lib, em._ctx)
    st, _ = bench.stage_profile(em._lib, em._ctx, lambda i, s: f_plain(i), min(a.steps, 10), with_stats=False)
    single_stage = {k: round(max(v - ev, 0.0), 5) for k, v in st.items()}
    rccl_world1_ms = None
    try:
        path = sharded.rccl_library_path().encode()
        uid = (ct.c_uint8 * 128)()
        if em._lib.emap_comm_unique_id(path, uid) == 0 and em._lib.emap_comm_init(em._ctx, path, uid, 0, 1) == 0:
            f_sh = frame_fn(em, True)
            for i in range(3):
                f_sh(i)
            rccl_world1_ms = timed_loops(em, f_sh, a.steps)
            em._lib.emap_comm_destroy(em._ctx)
    except Exception as ex:  # noqa: BLE001
        print("real RCCL at world size 1 unavailable: %s" % ex, file=sys.stderr)
    em.close()

    out = {"workload": bench.workload_text(ba, C, N, multimodal).replace("cfg5", a.workload).replace("cfg2", a.workload) + (
               "; WITH visibility clean-up + overlap clearance, strips of %s" % ("equal height" if a.equal_strips else "equal ray work") if a.rays else ""),
           "index_mode": mode, "steps": a.steps, "source_stamp": bench.source_stamp(),
           "method": "one MI355X; every frame is emap_update_sharded over the stream-ordered in-process RCCL stand-in (tests/fake_rccl/stream_rccl.hip); "
                     "solo = each rank alone (loop-back collectives: same bytes, streams and events), lockstep = all ranks alive as threads sharing "
                     "the GPU; times = median of 3 loops of K frames (device time on the strip's stream for solo, host wall for lockstep)",
           "single": {"frame_ms": round(single_ms, 5), "stage_ms_net": single_stage,
                      "frame_ms_sharded_world1_real_rccl": None if rccl_world1_ms is None else round(rccl_world1_ms, 5)},
           "splits": {}}

    out["ray_mode"] = ("by ray" if by_ray else "by row") if a.rays else None
    out["scene"] = a.scene
    out["cloud"] = "replicated to every rank" if (a.replicated_cloud or (a.rays and not by_ray)) else "bucketed per rank (emap_strip_point_mask: the points that can land in the rank's rows)"
    for G in [g for g in a.gs if g > 1]:
        # ---- solo: every rank alone, loop-back collectives ----------------------------------------------------------------------------
        # (not with rays by ray: a rank alone sees only its own rows in the all-reduced ray window -- every other cell reads as unknown,
        #  and the march would queue a visit for each of them; the lockstep run below has the real window)
        os.environ["STREAM_RCCL_LOOPBACK"] = "1"
        solo, stages, rows, shares = [], [], [], []
        for rank in range(G if not by_ray else 0):
            em = make_rank(G, rank, True)
            comm_init(em, new_uid(em._lib), rank, G)
            fr = frame_fn(em, True)
            warm(em, fr)
            solo.append(timed_loops(em, fr, a.steps))
            stg, _ = bench.stage_profile(em._lib, em._ctx, lambda i, s: fr(i), min(a.steps, 10), with_stats=False)
            stages.append({k: round(max(v - ev, 0.0), 5) for k, v in stg.items()})
            rows.append(em._rows01)
            shares.append(round(max(em._n_local) / float(N), 4))
            em._lib.emap_comm_destroy(em._ctx)
            drop_rank(em)
        os.environ["STREAM_RCCL_LOOPBACK"] = "0"
        sp = {}
        if solo:
            slow = int(np.argmax(solo))
            sp = {"rows": rows, "cloud_share_per_rank": shares, "stage_ms_net_per_rank": stages, "solo_frame_ms_per_rank": [round(v, 5) for v in solo], "solo_frame_ms_max": round(max(solo), 5),
                  "solo_frame_ms_sum": round(sum(solo), 5), "stage_ms_net_slowest_rank": stages[slow], "stage_ms_net_rank0": stages[0],
                  "hist_plus_scatter_ms_max": round(max(s["hist"] + s["scatter"] for s in stages), 5),
                  "speedup_solo_no_wire": round(single_ms / max(solo), 3)}
        # ---- lockstep: all ranks alive, collectives really meet ---------------------------------------------------------------------
        if not a.no_lockstep:
            ems = [make_rank(G, r, False) for r in range(G)]
            uid = new_uid(ems[0]._lib)
            bar = threading.Barrier(G + 1)
            errs, walls = [], [None] * 3

            def run(rank):
                try:
                    em = ems[rank]
                    comm_init(em, uid, rank, G)
                    fr = frame_fn(em, True)
                    warm(em, fr)
                    for rep in range(3):
                        em.sync(); bar.wait()
                        for i in range(a.steps):
                            fr(i)
                        em.sync(); bar.wait()
                    em._lib.emap_comm_destroy(em._ctx)
                except Exception as ex:  # noqa: BLE001
                    errs.append(ex); bar.abort()
            th = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(G)]
            [x.start() for x in th]
            try:
                for rep in range(3):
                    bar.wait(); t0 = time.perf_counter()
                    bar.wait(); walls[rep] = (time.perf_counter() - t0) * 1e3 / a.steps
            except threading.BrokenBarrierError:
                pass
            [x.join(timeout=300) for x in th]
            if errs:
                raise errs[0]
            for em in ems:
                drop_rank(em)
            lock = float(np.median(walls))
            sp.update({"lockstep_frame_ms_all_ranks_one_gpu": round(lock, 5), "work_inflation_vs_single": round(lock / single_ms, 3),
                       "lockstep_ms_per_rank_average": round(lock / G, 5), "speedup_bound_from_lockstep": round(G * single_ms / lock, 3)})
        out["splits"][str(G)] = sp

    # ---- the wire: cannot be measured on one GPU ------------------------------------------------------------------------------------
    wire = {"note": "an 8-rank all-reduce of 16 bytes and the halo rows over xGMI cannot be measured on one GPU; measured here: what the real "
                    "RCCL's world-1 frame adds to the plain frame (launch + kernel of the all-reduce: a lower bound of its multi-rank latency)",
            "real_rccl_world1_frame_minus_plain_ms": None if rccl_world1_ms is None else round(rccl_world1_ms - single_ms, 5),
            "projected_speedup": {}}
    for G in [g for g in a.gs if g > 1]:
        sp = out["splits"][str(G)]
        per_rank = sp.get("solo_frame_ms_max", sp.get("lockstep_ms_per_rank_average"))      # (rays by ray: the lockstep average -- uniform clouds give every rank N / G rays)
        if per_rank:
            wire["projected_speedup"][str(G)] = {"wire_%d_us" % us: round(single_ms / (per_rank + us * 1e-3), 3) for us in (0, 20, 60, 100)}
    out["wire"] = wire
    print(json.dumps(out))


if __name__ == "__main__":
    main()
