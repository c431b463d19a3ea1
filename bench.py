#!/usr/bin/env python3
"""bench.py -- Mpoints/s fused + map-update latency of the point-cloud hot path on MI355X.

A "step" is one frame of ``update_map_with_kernel`` (reference EM/elevation_mapping.py:316-391) over one
synthetic cloud that is already resident in HBM.  Default workload = BASELINE.json configs[1]:
1024x1024 map (cell_n incl. border), 1 M uniform-random points per frame, shipped core_param.yaml values,
visibility clean-up and overlap clearance off ("cfg2"); ``--workload cfg3`` turns both on.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3] [--points N] [--cell-n C]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (row strips, see sharded.py)

Prints ONE JSON line (rank 0): metric/value/unit + roofline + cpu_baseline objects (see DESIGN.md "Measurement").
"""
import argparse
import ctypes as ct
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg5"],
                    help="cfg2/cfg3: BASELINE configs[1]/[2] (1024^2, 1 M points); cfg5: 8192^2 multi-modal map "
                         "(height + RGB + 3 semantic layers), 16 M points, fp32 index mode, rays/overlap off")
    ap.add_argument("--points", type=int, default=None, help="default 1 M (cfg2/cfg3) or 16 M (cfg5)")
    ap.add_argument("--cell-n", type=int, default=None, help="default 1024 (cfg2/cfg3) or 8192 (cfg5)")
    ap.add_argument("--mode", default="reference_fp16", choices=["reference_fp16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-points", type=int, default=0, help="points of the CPU baseline sample (0 = auto)")
    ap.add_argument("--scatter", default="auto", choices=["auto", "atomic", "binned"])
    ap.add_argument("--sort-clouds", default="none", choices=["none", "tile", "angle"],
                    help="experiment: spatially coherent input order (real sensors deliver scan-ordered clouds)")
    ap.add_argument("--force-sharded", action="store_true", help="run the row-strip path even with one rank (self-test)")
    a = ap.parse_args()
    big = a.workload == "cfg5"
    a.cell_n = a.cell_n or (8192 if big else 1024)
    a.points = a.points or (16_000_000 if big else 1_000_000)
    return a


def workload_cfg(name):
    from elevation_mapping_cupy_amd.configs import CORE_PARAM_YAML
    cfg = dict(CORE_PARAM_YAML)
    if name == "cfg2":
        cfg.update(enable_visibility_cleanup=False, enable_overlap_clearance=False)
    return cfg


class Hip:
    """tiny ctypes view of the HIP runtime for device-resident input clouds (plumbing, not the product)."""

    def __init__(self):
        self.l = ct.CDLL("libamdhip64.so")

    def ck(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: hip error %d" % (what, rc))

    def set_device(self, d):
        self.ck(self.l.hipSetDevice(d), "hipSetDevice")

    def malloc(self, nbytes):
        p = ct.c_void_p()
        self.ck(self.l.hipMalloc(ct.byref(p), ct.c_size_t(nbytes)), "hipMalloc")
        return p

    def h2d(self, dst, arr):
        self.ck(self.l.hipMemcpy(dst, ct.c_void_p(arr.ctypes.data), ct.c_size_t(arr.nbytes), 1), "hipMemcpy")

    def sync(self):
        self.ck(self.l.hipDeviceSynchronize(), "hipDeviceSynchronize")


# Algorithmic bytes of each timed stage = the bytes that stage must move once (N points, L cells):
STAGE_BYTES = {
    "hist": lambda N, L: 12 * N + 16 * N,                    # xyz in, 16-B staging record out
    "scan": lambda N, L: 0,
    "scatter": lambda N, L: 16 * N + 16 * N,                 # staging record in, sorted record out (a pure permutation)
    "gate": lambda N, L: 16 * N + 16 * L,                    # per-tile error sums: sorted records + (h,v,valid,trav) of every cell, once
    "fuse": lambda N, L: 16 * N + 64 * L,                    # sorted records; cells read (staged per tile) + written once (fused average)
    "commit": lambda N, L: 40 * L + 64 * L,
    "rays": lambda N, L: 12 * N + 32 * L + 16 * L,           # cloud + map + ray accumulators once (the kernel is issue bound: see visits/s)
    "average": lambda N, L: 40 * L + 16 * L + 64 * L,
    "overlap": lambda N, L: 0,
    "post": lambda N, L: 32 * L + 4 * L + 4 * L + 12 * L,    # cells in; traversability_input, traversability, 3 normal planes out
}
STAGE_KERNEL = {"hist": "k_bin_hist", "scan": "k_bin_scan1", "scatter": "k_bin_scatter", "gate": "k_tile_count", "fuse": "k_tile_fuse",
                "commit": "k_commit", "rays": "k_rays<0, false", "average": "k_average", "overlap": "k_overlap", "post": "k_post"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 or world > 1 or a.force_sharded:
        from elevation_mapping_cupy_amd import sharded
        return sharded.bench_main(a, rank, world, local_rank)

    from elevation_mapping_cupy_amd import _lib
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    from elevation_mapping_cupy_amd.configs import parameter_from
    import _fixtures as fx

    cfg = workload_cfg(a.workload)
    C, N = a.cell_n, a.points
    multimodal = a.workload == "cfg5"
    if multimodal:
        a.mode = "fp32" if C > 2049 else a.mode
        cfg.update(enable_visibility_cleanup=False, enable_overlap_clearance=False)
    w = np.load(os.path.join(ROOT, "tests", "golden", "weights.npz"))
    weights = {k: w[k] for k in ("w1", "w2", "w3", "w_out")}
    par = parameter_from(cfg, C, a.mode, weights)
    par.device = local_rank
    emap = ElevationMap(par)
    emap.set_scatter_mode(a.scatter)
    lib, ctx = emap._lib, emap._ctx
    hip = Hip(); hip.set_device(local_rank)

    # 5 seeded clouds resident in HBM: x,y ~ U(-L/2, L/2), sensor-frame z ~ U(-.5,.5); timed clouds lowered
    NCLOUD = 2 if multimodal else 5
    if multimodal:   # x y z | rgb (packed 0x00RRGGBB) | 3 semantic features  -> colour + average fusions
        clouds_host = []
        for s_ in range(NCLOUD):
            p_ = fx.cloud(C, N, s_, dz=-0.02 * s_, extra=4)
            p_[:, 3] = np.random.default_rng(50 + s_).integers(0, 1 << 24, N, dtype=np.uint32).view(np.float32)
            clouds_host.append(p_)
        spec = _lib.EmapSemSpec()
        spec.n_col, spec.col_chan[0], spec.col_layer[0] = 1, 3, 0
        spec.n_sum = 3
        for k_ in range(3):
            spec.sum_chan[k_], spec.sum_layer[k_], spec.sum_kind[k_] = 4 + k_, 1 + k_, 0
        spec.alpha = 0.5
        if lib.emap_semantic_configure(ctx, 4):
            raise RuntimeError(lib.emap_last_error(ctx).decode())
    else:
        clouds_host = [fx.cloud(C, N, s, dz=(0.0 if s == 0 else -0.02 * s)) for s in range(NCLOUD)]
    if a.sort_clouds != "none":
        for k_, p_ in enumerate(clouds_host):
            if a.sort_clouds == "tile":
                ix = np.clip((p_[:, 0] / 0.04 + C / 2).astype(np.int64), 0, C - 1); iy = np.clip((p_[:, 1] / 0.04 + C / 2).astype(np.int64), 0, C - 1)
                key = (ix // 16) * (C // 64 + 1) * 4096 + (iy // 64) * 4096 + (ix % 16) * 64 + iy % 64
            else:
                key = np.arctan2(p_[:, 1], p_[:, 0])
            clouds_host[k_] = np.ascontiguousarray(p_[np.argsort(key, kind="stable")])
    stride = clouds_host[0].shape[1]
    clouds_dev = []
    for p in clouds_host:
        d = hip.malloc(p.nbytes); hip.h2d(d, p); clouds_dev.append(d)
    R = np.eye(3, dtype=np.float32).ravel().copy()
    t = np.array([0, 0, 1], np.float32)
    Rp, tp = _lib.f32p(R), _lib.f32p(t)

    def frame(i, stats=None):
        rc = lib.emap_set_points_device(ctx, clouds_dev[i % NCLOUD], ct.c_int64(N), ct.c_int64(stride))
        rc = rc or lib.emap_update(ctx, Rp, tp, ct.c_double(1.0), ct.c_double(1.0), stats)
        if multimodal:
            rc = rc or lib.emap_semantic_update(ctx, Rp, tp, ct.byref(spec))
        if rc:
            raise RuntimeError(lib.emap_last_error(ctx).decode())

    # map warm-up (3 frames + time ticks so that the ray pass has stale cells to act on)
    for i in range(3):
        frame(i)
        for _ in range(4):
            emap.update_time()
    emap.update_variance()
    for i in range(a.warmup):
        frame(i)
    emap.sync(); hip.sync()

    # ---- timed region: exactly K frames, sync on both sides -------------------------------------------------
    ms_dev = ct.c_float(0)
    t0 = time.perf_counter()
    lib.emap_timer_begin(ctx)
    for i in range(a.steps):
        frame(i)
    lib.emap_timer_end(ctx, ct.byref(ms_dev))
    emap.sync(); hip.sync()
    wall = time.perf_counter() - t0
    ms_per_step = wall * 1e3 / a.steps
    mpts = N * a.steps / wall / 1e6

    # ---- per-frame latency distribution (each frame individually synchronised) ---------------------------
    lat = []
    for i in range(min(a.steps, 40)):
        emap.sync()
        t1 = time.perf_counter(); frame(i); emap.sync(); lat.append((time.perf_counter() - t1) * 1e3)
    p10, p50, p90 = np.percentile(lat, [10, 50, 90])

    # ---- per-stage device time (hipEvents on the kernel's stream) -> roofline of the dominant kernel ------
    lib.emap_enable_stage_timing(ctx, 2)
    acc = np.zeros(10)
    reps = min(a.steps, 20)
    st = _lib.EmapStats()
    visits = 0
    for i in range(reps):
        frame(i, ct.byref(st))
        ms10 = (ct.c_float * 10)()
        lib.emap_get_stage_times(ctx, ms10)
        acc += np.array(list(ms10)); visits += st.ray_visits
    lib.emap_enable_stage_timing(ctx, 0)
    stage_ms = dict(zip(_lib.STAGES, (acc / reps).tolist()))
    # an event pair with NOTHING between its records is already ~4.5 us apart on this stack (marker processing); a stage interval
    # is that spacing + the kernel, so the spacing is calibrated and removed -- the result agrees with rocprofv3's kernel durations
    empty = []
    for _ in range(50):
        e_ms = ct.c_float(0)
        lib.emap_timer_begin(ctx); lib.emap_timer_end(ctx, ct.byref(e_ms)); empty.append(e_ms.value)
    ev_overhead = float(np.median(empty))
    L = C * C
    dom = max(stage_ms, key=stage_ms.get)
    dom_bytes = STAGE_BYTES[dom](N, L)
    dom_ms = max(stage_ms[dom] - ev_overhead, 1e-6)
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    frame_bytes = 12 * N + 56 * L + (16 * N + 32 * L if a.workload == 'cfg5' else 0)   # B_frame of BASELINE.md §5 (K = 0 extra channels, L = 0 semantic layers)
    # HBM traffic of the dominant kernel: from the committed rocprofv3 PMC passes of this same command
    # (tools/profile_round.sh -> profiles/pmc_<workload>.json; counters cannot be read from inside the process)
    traffic, traffic_src = None, None
    pmc_file = os.path.join(ROOT, "profiles", "pmc_%s.json" % a.workload)
    if os.path.exists(pmc_file) and C == 1024 and N == 1_000_000 and a.mode == "reference_fp16":
        kern = STAGE_KERNEL[dom]
        for name, rec in json.load(open(pmc_file))["kernels"].items():
            if name.startswith(kern):
                traffic, traffic_src = rec["hbm_bytes"], "profiles/pmc_%s.json (%s)" % (a.workload, name)
    roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes": dom_bytes, "kernel_ms": round(dom_ms, 5), "event_pair_overhead_ms": round(ev_overhead, 5),
            "stage_ms": {k: round(v, 5) for k, v in stage_ms.items()},   # raw event spacings (overhead included)
            "frame_algorithmic_bytes": frame_bytes,
            "frame_frac": round(frame_bytes / (ms_dev.value / a.steps * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "ray_visits_per_frame": int(visits / reps),
            "ray_visits_per_s": (round(visits / reps / (stage_ms["rays"] * 1e-3) / 1e9, 1) if visits else None), "visits_unit": "G cell visits/s"}

    # ---- CPU baseline: the oracle port, same workload, bounded sample ---------------------------------------
    cpu = None
    if not a.no_cpu_baseline and not multimodal:
        from oracle import emap_oracle as eo
        n_cpu = a.cpu_points or (N if a.workload == "cfg2" else min(N, 60000))
        P = eo.make_params(cfg, cell_n=C, mode=a.mode, weights=weights)
        def cpu_rate(threads):
            eo.set_threads(threads)
            om = eo.OracleMap(P)
            om.frame_c(clouds_host[0][:n_cpu], R, t, 1.0, 1.0)
            for _ in range(8):
                om.update_time()
            reps_cpu, t_cpu = 0, 0.0
            while reps_cpu < 5 and t_cpu < 8.0:
                s = time.perf_counter(); om.frame_c(clouds_host[(reps_cpu + 1) % NCLOUD][:n_cpu], R, t, 1.0, 1.0)
                t_cpu += time.perf_counter() - s; reps_cpu += 1
            eo.set_threads(1)
            return n_cpu * reps_cpu / t_cpu / 1e6, reps_cpu
        def usable_cores():
            n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            try:  # cgroup v2 CPU quota of the container
                q, per = open("/sys/fs/cgroup/cpu.max").read().split()
                if q != "max":
                    n = max(1, min(n, int(float(q) / float(per) + 0.5)))
            except (OSError, ValueError):
                pass
            return n
        avail = usable_cores()
        v1, reps_cpu = cpu_rate(1)
        best_v, best_n = v1, 1
        for nthr in sorted({min(avail, 8), min(avail, 32), avail} - {1}):    # oversubscription hurts: keep the best setting
            v, reps_cpu = cpu_rate(nthr)
            if v > best_v:
                best_v, best_n = v, nthr
        cpu = {"value": round(best_v, 4), "unit": "Mpoints/s", "cores": best_n, "kind": "port", "single_thread_value": round(v1, 4),
               "sample": "<=5 frames (<=8 s) of %d points on the %dx%d map, oracle/emap_oracle.c eo_frame (gcc -O2 -fopenmp); best of "
                         "1/8/32/%d threads, %d usable cores (os.cpu_count() = %d)" % (n_cpu, C, C, avail, avail, os.cpu_count() or 1)}

        # second CPU line: the reference's OWN kernel source compiled for the host (oracle/_ref, sequential, 1 thread) on a smaller
        # sample of the same clouds -- error_counting + add_points + average_map + dilation + normal kernels (its traversability
        # network is PyTorch and not part of that build)
        try:
            from oracle import build_ref, ref_kernels
            key = {(1024, "cfg2"): "yaml1024_norays", (1024, "cfg3"): "yaml1024", (202, "cfg2"): "yaml202_norays", (202, "cfg3"): "yaml202"}.get((C, a.workload))
            if key and a.mode == "reference_fp16" and ref_kernels.available(build_ref.PREBUILD[key]):
                rk = ref_kernels.RefKernels(build_ref.PREBUILD[key], build=False)
                n_ref = min(N, 200000 if a.workload == "cfg2" else 20000)
                m_ref = np.zeros((7, C, C), np.float32); m_ref[1] = cfg["initial_variance"]; m_ref[3] = 1
                nrm_ref = np.zeros((3, C, C), np.float32)
                Rf = np.ascontiguousarray(R, np.float32).ravel().copy(); tf = np.ascontiguousarray(t, np.float32)

                def ref_frame(k_):
                    p_ = np.ascontiguousarray(clouds_host[k_ % NCLOUD][:n_ref, :3])
                    nm_ = np.zeros((7, C, C), np.float32); e_ = np.zeros(1, np.float32); c_ = np.zeros(1, np.float32)
                    t0_ = time.perf_counter()
                    rk.error_counting(m_ref, p_, Rf, tf, nm_, e_, c_); rk.add_points(Rf, tf, nrm_ref, p_, m_ref, nm_); rk.average_map(nm_, m_ref)
                    dil_ = np.zeros((C, C), np.float32); dm_ = np.zeros((C, C), np.float32)
                    rk.dilation_filter(m_ref[5].copy(), (m_ref[2] + m_ref[6]).copy(), dil_, dm_)
                    rk.normal_filter(dil_, m_ref[2].copy(), nrm_ref)
                    return time.perf_counter() - t0_
                ref_frame(0); m_ref[4] += 1.0
                t_ref = [ref_frame(1), ref_frame(2)]
                cpu["reference_kernels"] = {"value": round(n_ref / float(np.mean(t_ref)) / 1e6, 4), "unit": "Mpoints/s", "cores": 1,
                                            "sample": "2 frames of %d points, the reference's kernel source compiled with g++ -O2 (oracle/build_ref.py)" % n_ref}
        except Exception as ex:  # noqa: BLE001 - the second line is optional
            print("reference-kernel CPU line skipped: %s" % ex, file=sys.stderr)

    out = {
        "metric": "Mpoints/s fused (map-update p50 latency in config)", "value": round(mpts, 2), "unit": "Mpoints/s",
        "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %dx%d map, %d uniform-random points/frame, core_param.yaml values, %s"
                               % (a.workload, C, C, N, "rays+overlap on" if a.workload == "cfg3" else
                                  ("height + RGB + 3 semantic layers, fp32 index mode" if multimodal else "add_points + variance fusion, rays/overlap off")),
                   "index_mode": a.mode, "latency_ms": {"p10": round(p10, 4), "p50": round(p50, 4), "p90": round(p90, 4)},
                   "device_ms_per_step": round(ms_dev.value / a.steps, 5), "cloud": "device resident (H2D excluded)"},
        "roofline": roof, "cpu_baseline": cpu,
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
