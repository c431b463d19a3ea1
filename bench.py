#!/usr/bin/env python3
"""bench.py -- Mpoints/s fused + map-update latency of the point-cloud hot path on MI355X.

A "step" is one frame of ``update_map_with_kernel`` (reference EM/elevation_mapping.py:316-391) over one
synthetic cloud that is already resident in HBM.  Default workload = BASELINE.json configs[1]:
1024x1024 map (cell_n incl. border), 1 M uniform-random points per frame, shipped core_param.yaml values,
visibility clean-up and overlap clearance off ("cfg2"); ``--workload cfg3`` turns both on.  The default cfg2 line also
carries ``config.cfg3``: the same process then times the cfg3 frame (rays + overlap) on a second map.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg5] [--points N] [--cell-n C]

``--gpus N`` (N > 1) runs N row strips, one process per GPU: started by an external launcher
(``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``: RANK / LOCAL_RANK / WORLD_SIZE in the
environment) or, when WORLD_SIZE is unset, by bench.py itself (elevation_mapping_cupy_amd/launch.py).  The strips talk
through the library's own RCCL communicator (no torch in the process); with fewer devices than ranks (single-GPU boxes)
the ranks share devices and the collectives fall back to torch.distributed/gloo.

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
PROFILE_ROUND = "r06"  # tag of the committed rocprofv3 summaries under profiles/ that this round's numbers refer to


def source_stamp():
    """sha256 over the kernel sources: every file under profiles/ is stamped with it (tools/profile_round.sh), and a profile is only
    quoted next to a live measurement when its stamp equals the stamp of the sources the loaded library was built from"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "elevation_mapping_cupy_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")) and not fn.startswith("emap_inpaint_"):      # (host-only translation units hold no kernel: a fix there must not stale the kernel profiles)
            h.update(fn.encode()); h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


def sem_in_frame(lib):
    """True when the multi-modal frames of this run declare their RGB / semantic fusion for the frame (emap_frame_semantics, round 6:
    fused inside emap_update's tile pass); False with an older library handed in for an A/B run (EMAP_HIP_LIB) or
    EMAP_BENCH_SEM_SEPARATE=1: the separate emap_semantic_update call behind every frame, as until round 5"""
    return hasattr(lib, "emap_frame_semantics") and os.environ.get("EMAP_BENCH_SEM_SEPARATE", "0") != "1"


def mm_frame(lib, ctx, Rp, tp, spec, stats=None, noise=1.0):
    """one multi-modal frame on the bound cloud: heights + RGB / semantic layers (returns the first non-zero status).  noise: position /
    orientation noise of the frame -- 1.0: above the YAML thresholds, the drift gate is OPEN (the per-tile drift statistics run); 0.0:
    the host can rule the gate out (elevation_mapping.py:346-349) and the frame skips them"""
    if sem_in_frame(lib):
        return lib.emap_frame_semantics(ctx, ct.byref(spec), 0) or lib.emap_update(ctx, Rp, tp, ct.c_double(noise), ct.c_double(noise), stats)
    return lib.emap_update(ctx, Rp, tp, ct.c_double(noise), ct.c_double(noise), stats) or lib.emap_semantic_update(ctx, Rp, tp, ct.byref(spec))


# VALU issue peak of the chip: 256 CUs x 4 SIMD-32 units, a wave64 vector instruction issues over 2 cycles at 2.4 GHz
# (MI355X_MICROARCH.md, "Wave scheduling") -> 1.2 G wave-instructions / s / SIMD.  (tools/clockbench.hip measured ~1 per ns and SIMD
# on dependent mixed code)
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0


def sq_valu_insts(workload, kernel_prefix):
    """SQ_INSTS_VALU per launch (wave instructions, average over the dispatches) of a kernel from the committed rocprofv3 --pmc SQ_*
    pass of this workload (profiles/<round>_<workload>_sq_counters.txt) -- (None, why) when the file is missing or was taken with
    other kernel sources"""
    f = os.path.join(ROOT, "profiles", "%s_%s_sq_counters.txt" % (PROFILE_ROUND, workload))
    if not os.path.exists(f):
        return None, "profiles/%s missing" % os.path.basename(f)
    lines = open(f).read().splitlines()
    stamp = next((ln.split(":", 1)[1].strip() for ln in lines if ln.startswith("# source_stamp:")), None)
    if stamp != source_stamp():
        return None, "profiles/%s: taken with other kernel sources (stamp %s, now %s)" % (os.path.basename(f), stamp, source_stamp())
    best, best_w = None, -1.0
    for ln in lines:
        name = ln.replace("void ", "", 1).lstrip()
        if name.startswith(kernel_prefix) and " SQ_INSTS_VALU " in ln:
            tail = ln.split("SQ_INSTS_VALU", 1)[1].split()
            v, n = float(tail[0]), float(tail[1].split("=")[1])
            if len(tail) > 2 and tail[2].startswith("med="):      # the median over the dispatches = the steady-state frames (the average includes warm-up / cold-start frames)
                v = float(tail[2].split("=")[1])
            if v * n > best_w:
                best, best_w = v, v * n
    return best, "profiles/%s" % os.path.basename(f)


def rocprof_kernel_us(workload, kernel_prefix):
    """median duration (us) of a kernel in profiles/<round>_<workload>_kernel_stats.txt -- None when the file is missing or was taken
    with other kernel sources (its '# source_stamp:' line)"""
    f = os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.txt" % (PROFILE_ROUND, workload))
    if not os.path.exists(f):
        return None, None
    lines = open(f).read().splitlines()
    stamp = next((ln.split(":", 1)[1].strip() for ln in lines if ln.startswith("# source_stamp:")), None)
    if stamp != source_stamp():
        return None, "profiles/%s: taken with other kernel sources (stamp %s, now %s)" % (os.path.basename(f), stamp, source_stamp())
    best, best_total = (None, None), -1.0
    for ln in lines:
        name = ln.replace("void ", "", 1).lstrip()
        if name.startswith(kernel_prefix):
            tok = ln[86:].split()
            try:     # columns: calls, avg_us, median_us, ...: the MEDIAN (the first launches after clear() / the warm-up's time ticks are not steady state)
                calls, avg, med = float(tok[0]), float(tok[1]), float(tok[2])
            except (IndexError, ValueError):
                continue
            if calls * avg > best_total:      # several template variants can match (202^2 frames, the counting variant of k_rays): the workload's own has the most time
                best_total, best = calls * avg, (med, "profiles/%s (median of %s launches; average %s us)" % (os.path.basename(f), tok[0], tok[1]))
    return best


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5", "ref_main"],
                    help="ref_main: the reference's own profiling loop (EM/elevation_mapping.py:925-967) through the drop-in package, per-call "
                         "host latencies + one stage-timed line each for the MinFilter plugin, the layer read-back and the camera path; cfg2/cfg3: BASELINE configs[1]/[2] (1024^2, 1 M points); cfg4: configs[3] (4096^2, 4 M points, fp32 index mode, "
                         "rays + overlap on); cfg5: 8192^2 multi-modal map (height + RGB + 3 semantic layers), 16 M points, fp32 index "
                         "mode, rays/overlap off")
    ap.add_argument("--points", type=int, default=None, help="default 1 M (cfg2/cfg3), 4 M (cfg4) or 16 M (cfg5)")
    ap.add_argument("--cell-n", type=int, default=None, help="default 1024 (cfg2/cfg3), 4096 (cfg4) or 8192 (cfg5)")
    ap.add_argument("--mode", default="reference_fp16", choices=["reference_fp16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cfg3", action="store_true", help="skip the config.cfg3 and config.cfg1 sub-measurements of the default cfg2 line")
    ap.add_argument("--interleaved-cloud", action="store_true", help="cfg5: bind the multi-modal cloud as interleaved (N, 3 + K) device rows instead of "
                    "the de-interleaved layout an uploaded cloud has (emap_upload_points)")
    ap.add_argument("--no-terrain", action="store_true", help="skip config.cfg3.terrain (the cfg3 frame on a spatially coherent, scan-ordered scene)")
    ap.add_argument("--no-large", action="store_true", help="skip the config.cfg4 / config.cfg5 sub-measurements (4096^2 and 8192^2 on one GPU) of the default line")
    ap.add_argument("--cpu-points", type=int, default=0, help="points of the CPU baseline sample (0 = auto)")
    ap.add_argument("--scatter", default="auto", choices=["auto", "atomic", "binned"])
    ap.add_argument("--sort-clouds", default="none", choices=["none", "tile", "angle"],
                    help="experiment: spatially coherent input order (real sensors deliver scan-ordered clouds)")
    ap.add_argument("--force-sharded", action="store_true", help="run the row-strip path even with one rank (self-test)")
    ap.add_argument("--pre-shift", type=int, nargs=2, default=None, metavar=("ROWS", "COLS"),
                    help="experiment: shift the map by this many cells before the frames (circular origin off the tile grid)")
    ap.add_argument("--gate", default="open", choices=["open", "shut"],
                    help="drift gate of the timed frames: open = position / orientation noise 1.0, above the YAML thresholds (BASELINE's frame: "
                         "the drift statistics run and the gate fires); shut = noise 0.0, the host rules the gate out and the frame skips the statistics")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous plumbing only, no GPU work (CPU test hook)")
    a = ap.parse_args(argv)
    a.cell_n = a.cell_n or {"cfg5": 8192, "cfg4": 4096}.get(a.workload, 1024)
    a.points = a.points or {"cfg5": 16_000_000, "cfg4": 4_000_000}.get(a.workload, 1_000_000)
    if a.cell_n > 2049:
        a.mode = "fp32"              # the reference's half-precision helpers cannot address more than 2049 cells per axis
    return a


def workload_cfg(name):
    from elevation_mapping_cupy_amd.configs import CORE_PARAM_YAML
    cfg = dict(CORE_PARAM_YAML)
    if name in ("cfg2", "cfg5"):
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

    def free(self, p):
        self.l.hipFree(p)

    def h2d(self, dst, arr):
        self.ck(self.l.hipMemcpy(dst, ct.c_void_p(arr.ctypes.data), ct.c_size_t(arr.nbytes), 1), "hipMemcpy")

    def sync(self):
        self.ck(self.l.hipDeviceSynchronize(), "hipDeviceSynchronize")

    def pinned_h2d_gbs(self, nbytes, reps=5):
        """bandwidth of a plain hipMemcpy from pinned host memory (the PCIe-bound rate the inclusive figures are compared with)"""
        h, d = ct.c_void_p(), self.malloc(nbytes)
        self.ck(self.l.hipHostMalloc(ct.byref(h), ct.c_size_t(nbytes), 0), "hipHostMalloc")
        ct.memset(h, 1, nbytes)
        best = 0.0
        for _ in range(reps):
            self.sync(); t0 = time.perf_counter()
            self.ck(self.l.hipMemcpy(d, h, ct.c_size_t(nbytes), 1), "hipMemcpy"); self.sync()
            best = max(best, nbytes / (time.perf_counter() - t0) / 1e9)
        self.l.hipHostFree(h); self.l.hipFree(d)
        return best


# Algorithmic bytes of each timed stage = the bytes that stage must move once WHATEVER the data (N points, L cells) -- the kernels
# as they are since round 3: the histogram pass only reads the cloud, the scatter pass reads it again and writes the 16-byte sorted
# records, the tile kernels read the records and the 16-byte hot half cells (the fuse kernel writes them back; the cold halves it
# touches depend on the data -- fused cells, stale cells in front of a ray pass -- and are NOT counted: a lower bound).  "post" follows
# dilation 12 B/cell (2 planes in, 1 out) + normal filter 20 B/cell + traversability filter 8 B/cell.
STAGE_BYTES = {
    "hist": lambda N, L: 12 * N,                             # xyz in (the per-block histogram rows are 4 B x tiles x blocks: < 2 %)
    "scan": lambda N, L: 0,
    "scatter": lambda N, L: 12 * N + 16 * N,                 # xyz in again (the geometry is recomputed), sorted record out
    "gate": lambda N, L: 16 * N + 16 * L,                    # per-tile error sums: sorted records + (h,v,valid,trav) of every cell, once
    "fuse": lambda N, L: 16 * N + 32 * L,                    # sorted records; hot half cells read (staged per tile) and written once
    "commit": lambda N, L: 40 * L + 64 * L,
    "rays": lambda N, L: 12 * N + 32 * L + 16 * L,           # cloud + map + ray accumulators once (the kernel is issue bound: see visits/s)
    "average": lambda N, L: 40 * L + 16 * L + 64 * L,
    "overlap": lambda N, L: 0,
    "post": lambda N, L: 40 * L,
}
STAGE_KERNEL = {"hist": "k_bin_hist", "scan": "k_bin_scan", "scatter": "k_bin_scatter", "gate": "k_tile_count", "fuse": "k_tile_fuse",
                "commit": "k_commit", "rays": "k_rays<", "average": "k_ray_apply", "overlap": "k_overlap", "post": "k_post"}


def load_weights():
    w = np.load(os.path.join(ROOT, "tests", "golden", "weights.npz"))
    return {k: w[k] for k in ("w1", "w2", "w3", "w_out")}


def host_clouds(a, C, N, multimodal):
    """the seeded clouds of SURVEY §8d: x,y ~ U(-L/2, L/2), sensor-frame z ~ U(-.5,.5); the timed clouds are lowered"""
    import _fixtures as fx
    ncloud = 2 if multimodal else 5
    if multimodal:   # x y z | rgb (packed 0x00RRGGBB) | 3 semantic features  -> colour + average fusions
        clouds = []
        for s_ in range(ncloud):
            p_ = fx.cloud(C, N, s_, dz=-0.02 * s_, extra=4)
            p_[:, 3] = np.random.default_rng(50 + s_).integers(0, 1 << 24, N, dtype=np.uint32).view(np.float32)
            clouds.append(p_)
    else:
        clouds = [fx.cloud(C, N, s, dz=(0.0 if s == 0 else -0.02 * s)) for s in range(ncloud)]
    if a.sort_clouds != "none":
        for k_, p_ in enumerate(clouds):
            if a.sort_clouds == "tile":
                ix = np.clip((p_[:, 0] / 0.04 + C / 2).astype(np.int64), 0, C - 1); iy = np.clip((p_[:, 1] / 0.04 + C / 2).astype(np.int64), 0, C - 1)
                key = (ix // 16) * (C // 64 + 1) * 4096 + (iy // 64) * 4096 + (ix % 16) * 64 + iy % 64
            else:
                key = np.arctan2(p_[:, 1], p_[:, 0])
            clouds[k_] = np.ascontiguousarray(p_[np.argsort(key, kind="stable")])
    return clouds


def device_clouds(hip, clouds_host, split):
    """device-resident copies of the clouds.  Clouds with extra channels are bound DE-INTERLEAVED by default -- an (N, 3) xyz matrix and
    an (N, K) channel matrix, the layout emap_upload_points (input_pointcloud) gives every uploaded cloud -- or, with split = False,
    as the interleaved (N, 3 + K) rows update_map_with_kernel takes from a caller that already holds a device array"""
    out = []
    for p in clouds_host:
        K = p.shape[1] - 3
        if split and K > 0:
            xyz, ch = np.ascontiguousarray(p[:, :3]), np.ascontiguousarray(p[:, 3:])
            dx = hip.malloc(xyz.nbytes); hip.h2d(dx, xyz)
            dc = hip.malloc(ch.nbytes); hip.h2d(dc, ch)
            out.append(("split", dx, dc, K))
        else:
            d = hip.malloc(p.nbytes); hip.h2d(d, p)
            out.append(("rows", d, None, p.shape[1]))
    return out


def bind_cloud(lib, ctx, cl, N):
    if cl[0] == "split":
        return lib.emap_set_points_device_split(ctx, cl[1], cl[2], ct.c_int64(N), ct.c_int64(cl[3]))
    return lib.emap_set_points_device(ctx, cl[1], ct.c_int64(N), ct.c_int64(cl[3]))


def free_clouds(hip, clouds):
    for cl in clouds:
        hip.free(cl[1])
        if cl[2] is not None:
            hip.free(cl[2])


def event_overhead(lib, ctx):
    """an event pair with NOTHING between its records is already ~4.5 us apart on this stack (marker processing); a stage interval
    is that spacing + the kernel, so the spacing is calibrated and removed -- the result agrees with rocprofv3's kernel durations"""
    empty = []
    for _ in range(50):
        e_ms = ct.c_float(0)
        lib.emap_timer_begin(ctx); lib.emap_timer_end(ctx, ct.byref(e_ms)); empty.append(e_ms.value)
    return float(np.median(empty))


def stage_profile(lib, ctx, frame, reps, with_stats=True):
    """HIP-event spacings of the stages, measured on the SAME kernels the timed frames run; the cell-visit count of the visibility
    pass comes from two extra frames with the counting variant of k_rays (a few more vector instructions per step)"""
    from elevation_mapping_cupy_amd import _lib
    lib.emap_enable_stage_timing(ctx, 1)
    acc = np.zeros(10)
    st = _lib.EmapStats()
    for i in range(reps):
        frame(i, None)
        ms10 = (ct.c_float * 10)()
        lib.emap_get_stage_times(ctx, ms10)
        acc += np.array(list(ms10))
    visits = 0
    if with_stats:
        lib.emap_enable_stage_timing(ctx, 2)
        for i in range(2):
            frame(reps + i, ct.byref(st))
            visits += st.ray_visits / 2
    lib.emap_enable_stage_timing(ctx, 0)
    return dict(zip(_lib.STAGES, (acc / reps).tolist())), visits


def cold_start(em, frame, n=10):
    """the visibility pass on an UNKNOWN map: clear(), then n frames with per-stage events -- every visit of an unknown cell min-reduces
    its upper bound, so the first frames are the pass's worst case (k_rays event spacing per frame, its maximum and the steady median)"""
    lib, ctx = em._lib, em._ctx
    em.clear()
    lib.emap_enable_stage_timing(ctx, 1)
    rays = []
    for i in range(n):
        frame(i, None)
        ms10 = (ct.c_float * 10)()
        lib.emap_get_stage_times(ctx, ms10)
        rays.append(float(ms10[6]))
    lib.emap_enable_stage_timing(ctx, 0)
    return {"rays_ms_per_frame": [round(x, 4) for x in rays], "max": round(max(rays), 4), "median_last5": round(float(np.median(rays[-5:])), 4),
            "max_over_median": round(max(rays) / max(float(np.median(rays[-5:])), 1e-9), 2)}


def scene_change(em, frame_a, frame_b, reps=3):
    """the first frames of scene B after frames of scene A (device time per frame = sum of the stage spacings): a cloud with heavy
    sort tiles right after clouds without -- the host sizes the extra workgroups of the tile kernels by the last frame it has heard
    of, so the first frame of the new scene only finds the standing pool (emap_api.hip: emap_count)"""
    lib, ctx = em._lib, em._ctx
    lib.emap_enable_stage_timing(ctx, 1)
    ms10 = (ct.c_float * 10)()
    rows = []
    for rep in range(reps):
        for i in range(3):
            frame_a(i, None)
        em.sync()
        row = []
        for i in range(4):
            frame_b(rep + i, None)
            lib.emap_get_stage_times(ctx, ms10)
            row.append((float(sum(ms10)), float(ms10[3]), float(ms10[4])))
        rows.append(row)
    lib.emap_enable_stage_timing(ctx, 0)
    med = lambda k, j: round(float(np.median([r[k][j] for r in rows])), 4)
    return {"first_frame_ms": med(0, 0), "second_frame_ms": med(1, 0), "fourth_frame_ms": med(3, 0),
            "first_over_fourth": round(med(0, 0) / max(med(3, 0), 1e-9), 3),
            "first_frame_gate_fuse_ms": [med(0, 1), med(0, 2)], "fourth_frame_gate_fuse_ms": [med(3, 1), med(3, 2)]}


def ray_samples(cloud, cfg):
    """~ samples one frame's rays take (custom_kernels.py:203-211: one every resolution / sqrt(2) up to min(|point - sensor|,
    max_ray_length)) -- the synthetic clouds are given in the sensor frame; every row counted (invalid points are a small share)"""
    r = np.minimum(np.linalg.norm(np.asarray(cloud[:, :3], np.float64), axis=1), float(cfg["max_ray_length"]))
    return int(np.floor(r[np.isfinite(r)] / (float(cfg["resolution"]) / 2 ** 0.5)).sum())


def roofline(stage_ms, ev_overhead, N, L, workload, frame_bytes, dev_ms_per_step, visits, pmc_ok, stage_bytes=None, samples=None):
    sb = stage_bytes or {k: f(N, L) for k, f in STAGE_BYTES.items()}
    cand = {k: v for k, v in stage_ms.items() if sb[k] > 0}
    dom = max(cand, key=cand.get)
    dom_bytes = sb[dom]
    # the kernel's duration = the spacing of the two HIP events around it on its stream.  That is what rocprofv3 reports as the kernel's
    # duration too (profiles/rNN_*_kernel_stats.txt: both include the ~4 us every launch occupies the stream); `kernel_ms_net` takes
    # the spacing of an EMPTY event pair off (the pure execution time, used in DESIGN.md's ablations)
    dom_ms = max(stage_ms[dom], 1e-6)
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    # HBM traffic of the dominant kernel: from the committed rocprofv3 PMC passes of this same command
    # (tools/profile_round.sh -> profiles/pmc_<workload>.json; counters cannot be read from inside the process)
    traffic, traffic_src = None, None
    pmc_file = os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (PROFILE_ROUND, workload))
    if pmc_ok and os.path.exists(pmc_file):
        kern = STAGE_KERNEL[dom]
        pj = json.load(open(pmc_file))
        if pj.get("source_stamp") == source_stamp():
            most = -1.0
            for name, rec in pj["kernels"].items():       # several template variants can match (the default run also times the 202^2 map): the workload's own moved the most bytes in total
                if name.startswith(kern) and rec["hbm_bytes"] * float(rec.get("launches", 1)) > most:
                    most = rec["hbm_bytes"] * float(rec.get("launches", 1))
                    traffic, traffic_src = rec["hbm_bytes"], "profiles/%s (%s, median of %d launches)" % (os.path.basename(pmc_file), name, rec.get("launches", 0))
    # `frac` is LIVE: the dominant kernel's algorithmic bytes over its event spacing in THIS run.  Beside it, when profiles/ holds a
    # rocprofv3 summary of this command taken with THESE kernel sources, the same bytes over that summary's median duration
    kus, ksrc = rocprof_kernel_us(workload, STAGE_KERNEL[dom]) if pmc_ok else (None, None)
    frac_rocprof = round(dom_bytes / (kus * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if kus else None
    if dom == "rays":
        # The visibility march is bound by vector-instruction ISSUE, not by HBM (12 N + 48 L bytes against ~20 instructions for each of
        # ~3.5e8 samples): its roofline is SQ_INSTS_VALU per launch (committed SQ-counter pass of this command, same kernel sources) over
        # the live kernel time, against the rate the chip's 1024 SIMDs can issue wave64 vector instructions at.  The HBM figure stays
        # beside it as `hbm`.
        insts, isrc = sq_valu_insts(workload, "k_rays<") if pmc_ok else (None, "no SQ-counter pass for this map / cloud size")
        ginst = insts / (dom_ms * 1e-3) / 1e9 if insts else None
        return {"bound": "valu", "kernel": dom, "achieved": round(ginst, 1) if ginst else None, "peak": round(VALU_PEAK_GINST, 1),
                "unit": "G wave-instructions/s", "frac": round(ginst / VALU_PEAK_GINST, 4) if ginst else None,
                "frac_source": "SQ_INSTS_VALU per launch (median over the dispatches, %s) / live hipEvent spacing of the kernel (kernel_ms) / (1024 SIMDs x 1.2 G wave-instructions/s)" % isrc,
                "valu_insts_per_launch": insts,
                "hbm": {"achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "frac_rocprof": frac_rocprof, "algorithmic_bytes": int(dom_bytes)},
                "kernel_us_rocprof": kus, "kernel_us_rocprof_source": ksrc,
                "source_stamp": source_stamp(), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes": int(dom_bytes), "kernel_ms": round(dom_ms, 5), "kernel_ms_net": round(max(dom_ms - ev_overhead, 0.0), 5),
                "event_pair_overhead_ms": round(ev_overhead, 5),
                "stage_ms": {k: round(v, 5) for k, v in stage_ms.items()},
                "frame_algorithmic_bytes": int(frame_bytes),
                "frame_frac": round(frame_bytes / (dev_ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "ray_visits_per_frame": int(visits),
                "ray_visits_per_s": (round(visits / (stage_ms["rays"] * 1e-3) / 1e9, 1) if visits else None), "visits_unit": "G cell visits/s",
                "ray_samples_per_frame": samples, "ray_samples_per_s": (round(samples / (stage_ms["rays"] * 1e-3) / 1e9, 1) if samples else None),
                "valu_insts_per_sample": (round(insts * 64.0 / samples, 1) if insts and samples else None), "samples_unit": "G samples/s (a sample = one step of one ray; ~, from the cloud)"}
    return {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_source": "live: hipEvent spacing of the kernel on its stream in this run (kernel_ms)",
            "frac_rocprof": frac_rocprof, "kernel_us_rocprof": kus, "kernel_us_rocprof_source": ksrc,
            "source_stamp": source_stamp(), "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes": int(dom_bytes), "kernel_ms": round(dom_ms, 5), "kernel_ms_net": round(max(dom_ms - ev_overhead, 0.0), 5),
            "event_pair_overhead_ms": round(ev_overhead, 5),
            "stage_ms": {k: round(v, 5) for k, v in stage_ms.items()},   # raw event spacings (overhead included)
            "frame_algorithmic_bytes": int(frame_bytes),
            "frame_frac": round(frame_bytes / (dev_ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "ray_visits_per_frame": int(visits),
            "ray_visits_per_s": (round(visits / (stage_ms["rays"] * 1e-3) / 1e9, 1) if visits else None), "visits_unit": "G cell visits/s"}


def cpu_baseline(a, cfg, C, N, clouds_host, weights, R, t):
    """the oracle port (and the reference's own compiled kernels) on a bounded sample of the same workload"""
    from oracle import emap_oracle as eo
    ncloud = len(clouds_host)
    n_cpu = a.cpu_points or (N if a.workload == "cfg2" else min(N, 60000))
    P = eo.make_params(cfg, cell_n=C, mode=a.mode, weights=weights)

    def cpu_rate(threads):
        eo.set_threads(threads)
        om = eo.OracleMap(P)
        om.frame_c(clouds_host[0][:n_cpu, :3], R, t, 1.0, 1.0)
        for _ in range(8):
            om.update_time()
        reps_cpu, t_cpu = 0, 0.0
        while reps_cpu < 5 and t_cpu < 8.0:
            s = time.perf_counter(); om.frame_c(clouds_host[(reps_cpu + 1) % ncloud][:n_cpu, :3], R, t, 1.0, 1.0)
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
                p_ = np.ascontiguousarray(clouds_host[k_ % ncloud][:n_ref, :3])
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
    return cpu


def workload_text(a, C, N, multimodal):
    return "%s: %dx%d map, %d uniform-random points/frame, core_param.yaml values, %s" % (
        a.workload, C, C, N, "rays+overlap on" if a.workload in ("cfg3", "cfg4") else
        ("height + RGB + 3 semantic layers, fp32 index mode" if multimodal else "add_points + variance fusion, rays/overlap off")) + (
        "; drift gate SHUT (noise 0.0: the host rules the gate out)" if getattr(a, "gate", "open") == "shut" else "")


# -------------------------------------------------------------------------------------------------------------------------------
def run_single(a, local_rank=0):
    from elevation_mapping_cupy_amd import _lib
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    from elevation_mapping_cupy_amd.configs import parameter_from

    cfg = workload_cfg(a.workload)
    C, N = a.cell_n, a.points
    multimodal = a.workload == "cfg5"
    if multimodal and C > 2049:
        a.mode = "fp32"
    weights = load_weights()
    par = parameter_from(cfg, C, a.mode, weights)
    par.device = local_rank
    emap = ElevationMap(par)
    emap.set_scatter_mode(a.scatter)
    lib, ctx = emap._lib, emap._ctx
    hip = Hip(); hip.set_device(local_rank)

    clouds_host = host_clouds(a, C, N, multimodal)
    NCLOUD = len(clouds_host)
    spec = None
    if multimodal:
        spec = _lib.EmapSemSpec()
        spec.n_col, spec.col_chan[0], spec.col_layer[0] = 1, 3, 0
        spec.n_sum = 3
        for k_ in range(3):
            spec.sum_chan[k_], spec.sum_layer[k_], spec.sum_kind[k_] = 4 + k_, 1 + k_, 0
        spec.alpha = 0.5
        if lib.emap_semantic_configure(ctx, 4):
            raise RuntimeError(lib.emap_last_error(ctx).decode())
    clouds_dev = device_clouds(hip, clouds_host, not a.interleaved_cloud)
    R = np.eye(3, dtype=np.float32).ravel().copy()
    t = np.array([0, 0, 1], np.float32)
    Rp, tp = _lib.f32p(R), _lib.f32p(t)

    noise = 0.0 if a.gate == "shut" else 1.0

    def make_frame(lib, ctx, cl_dev=None, n_pts=None):
        cl_dev = cl_dev or clouds_dev
        n_pts = n_pts or N

        def frame(i, stats=None):
            rc = bind_cloud(lib, ctx, cl_dev[i % len(cl_dev)], n_pts)
            if multimodal:
                rc = rc or mm_frame(lib, ctx, Rp, tp, spec, stats, noise)
            else:
                rc = rc or lib.emap_update(ctx, Rp, tp, ct.c_double(noise), ct.c_double(noise), stats)
            if rc:
                raise RuntimeError(lib.emap_last_error(ctx).decode())
        return frame
    frame = make_frame(lib, ctx)

    def warm(em, fr):
        if a.pre_shift:
            em.shift_map_xy(np.array(a.pre_shift))
        # map warm-up (3 frames + time ticks so that the ray pass has stale cells to act on)
        for i in range(3):
            fr(i)
            for _ in range(4):
                em.update_time()
        em.update_variance()
        for i in range(a.warmup):
            fr(i)
        em.sync(); hip.sync()

    def timed(em, fr, steps, loops=5):
        """`loops` timed loops of exactly `steps` frames each, device sync on both sides of every loop; returns the MEDIAN loop
        (host wall seconds, device ms on the context's stream) and every loop's wall time"""
        res = []
        for _ in range(loops):
            ms_dev = ct.c_float(0)
            em.sync(); hip.sync()
            t0 = time.perf_counter()
            em._lib.emap_timer_begin(em._ctx)
            for i in range(steps):
                fr(i)
            em._lib.emap_timer_end(em._ctx, ct.byref(ms_dev))
            em.sync(); hip.sync()
            res.append((time.perf_counter() - t0, ms_dev.value))
        med = sorted(res)[len(res) // 2]
        return med[0], med[1], [round(w * 1e3 / steps, 5) for w, _ in res]

    def latencies(em, fr, n):
        lat = []
        for i in range(n):
            em.sync()
            t1 = time.perf_counter(); fr(i); em.sync(); lat.append((time.perf_counter() - t1) * 1e3)
        return [float(x) for x in np.percentile(lat, [10, 50, 90])]

    warm(emap, frame)
    # ---- timed region: exactly K frames, sync on both sides -------------------------------------------------
    wall, ms_dev, loops_ms = timed(emap, frame, a.steps)      # `value` = the median of five timed loops of K frames
    ms_per_step = wall * 1e3 / a.steps
    mpts = N * a.steps / wall / 1e6
    # ---- per-frame latency distribution (each frame individually synchronised) ---------------------------
    p10, p50, p90 = latencies(emap, frame, min(a.steps, 40))
    # ---- per-stage device time (hipEvents on the kernel's stream) -> roofline of the dominant kernel ------
    stage_ms, visits = stage_profile(lib, ctx, frame, min(a.steps, 20))
    ev_overhead = event_overhead(lib, ctx)
    L = C * C
    frame_bytes = 12 * N + 56 * L + (16 * N + 32 * L if multimodal else 0)   # B_frame of BASELINE.md §5 (K extra channels, L semantic layers)
    pmc_ok = C == 1024 and N == 1_000_000 and a.mode == "reference_fp16"
    samples = ray_samples(clouds_host[0], cfg) if cfg.get("enable_visibility_cleanup") else None
    roof = roofline(stage_ms, ev_overhead, N, L, a.workload, frame_bytes, ms_dev / a.steps, visits, pmc_ok, samples=samples)

    # ---- config.cfg3: the frame WITH the visibility pass + overlap clearance, same process, second map ------------------
    cfg3 = None
    if a.workload == "cfg2" and not a.no_cfg3 and not multimodal:
        cfgr = workload_cfg("cfg3")
        par3 = parameter_from(cfgr, C, a.mode, weights); par3.device = local_rank
        em3 = ElevationMap(par3); em3.set_scatter_mode(a.scatter)
        fr3 = make_frame(em3._lib, em3._ctx)
        warm(em3, fr3)
        k3 = max(3, min(a.steps, 20))
        wall3, ms3, _ = timed(em3, fr3, k3, loops=3)
        lat3 = latencies(em3, fr3, min(k3, 10))
        st3, vis3 = stage_profile(em3._lib, em3._ctx, fr3, min(k3, 8))
        cold = cold_start(em3, fr3)
        r3 = roofline(st3, ev_overhead, N, L, "cfg3", frame_bytes, ms3 / k3, vis3, pmc_ok, samples=ray_samples(clouds_host[0], cfgr))
        cfg3 = {"workload": "cfg3: same map and clouds with enable_visibility_cleanup + enable_overlap_clearance",
                "value": round(N * k3 / wall3 / 1e6, 2), "unit": "Mpoints/s", "steps": k3, "ms_per_step": round(wall3 * 1e3 / k3, 5),
                "latency_ms": {"p10": round(lat3[0], 4), "p50": round(lat3[1], 4), "p90": round(lat3[2], 4)},
                "dominant_kernel": r3["kernel"], "kernel_ms": r3["kernel_ms"], "bound": r3["bound"], "frac": r3["frac"], "unit": r3["unit"], "achieved": r3["achieved"],
                "hbm": r3.get("hbm"), "traffic": r3["traffic"],
                "ray_visits_per_frame": r3["ray_visits_per_frame"], "ray_visits_per_s": r3["ray_visits_per_s"],
                "ray_samples_per_frame": r3.get("ray_samples_per_frame"), "ray_samples_per_s": r3.get("ray_samples_per_s"), "valu_insts_per_sample": r3.get("valu_insts_per_sample"),
                "stage_ms": r3["stage_ms"], "cold_start_ms": cold}
        em3.close()
        # ---- the same frame on a spatially COHERENT scene: scan-ordered beams ray-cast at a terrain with walls (tests/_fixtures.py:
        # terrain_cloud) instead of white noise -- neighbouring beams end in neighbouring cells and travel through the same cells,
        # rays do not dive under the ground they measured, 11 % of the cells are ever seen (density ~ 1/range).  Sensor noise and a
        # scene that keeps changing (every second obstacle moves between the clouds) as a robot would see it.
        if C == 1024 and N == 1_000_000 and not a.no_terrain:
            import _fixtures as fx
            th = [fx.terrain_cloud(C, 2000, 500, s_, shift=sh) for s_, sh in enumerate((0.0, 0.4, -0.3, 0.2))]
            td = device_clouds(hip, th, True)
            emt = ElevationMap(par3); emt.set_scatter_mode(a.scatter)
            frt = make_frame(emt._lib, emt._ctx, td, th[0].shape[0])
            warm(emt, frt)
            wallt, mst, _ = timed(emt, frt, k3, loops=3)
            stt, vist = stage_profile(emt._lib, emt._ctx, frt, min(k3, 8))
            valid_t = float((emt.get_layer_raw(2) > 0.5).mean())
            chg = scene_change(emt, make_frame(emt._lib, emt._ctx), frt)      # uniform clouds (no heavy tiles), then this scene
            coldt = cold_start(emt, frt)
            cfg3["terrain"] = {"workload": "cfg3 on a coherent scene: 2000 x 500 scan-ordered beams ray-cast at rolling ground with moving walls, 1 cm range noise",
                               "ms_per_step": round(wallt * 1e3 / k3, 5), "value": round(th[0].shape[0] * k3 / wallt / 1e6, 2), "unit": "Mpoints/s",
                               "valid_cell_fraction": round(valid_t, 4), "ray_visits_per_frame": int(vist),
                               "stage_ms": {k_: round(v_, 5) for k_, v_ in stt.items() if v_ > 0}, "cold_start_ms": coldt,
                               "after_uniform_frames": chg}
            emt.close()
            free_clouds(hip, td)

    # ---- config.cfg1: robot scale -- the 200 x 200 (+ border) map and 50 k-point clouds the reference ships with (BASELINE configs[0]
    # on the GPU), without and with the visibility pass: per-frame latency is what a user of the drop-in sees first
    cfg1 = None
    if a.workload == "cfg2" and not a.no_cfg3 and not multimodal:
        C1, N1 = 202, 50_000
        h1 = host_clouds(a, C1, N1, False)
        d1 = []
        for p_ in h1:
            dd = hip.malloc(p_.nbytes); hip.h2d(dd, p_); d1.append(dd)
        cfg1 = {"workload": "cfg1: %dx%d map, %d uniform-random points/frame, core_param.yaml values, device-resident clouds" % (C1, C1, N1)}
        for tag, wl in (("fusion_only", "cfg2"), ("rays_overlap", "cfg3")):
            par1 = parameter_from(workload_cfg(wl), C1, a.mode if C1 <= 2049 else "fp32", weights); par1.device = local_rank
            em1 = ElevationMap(par1)
            l1, c1 = em1._lib, em1._ctx

            def fr1(i, stats=None, l1=l1, c1=c1):
                rc = l1.emap_set_points_device(c1, d1[i % len(d1)], ct.c_int64(N1), ct.c_int64(3))
                rc = rc or l1.emap_update(c1, Rp, tp, ct.c_double(1.0), ct.c_double(1.0), stats)
                if rc:
                    raise RuntimeError(l1.emap_last_error(c1).decode())
            warm(em1, fr1)
            k1 = max(20, min(a.steps * 4, 200))
            wall1, _, _ = timed(em1, fr1, k1, loops=3)
            lat1 = latencies(em1, fr1, 40)
            cfg1[tag] = {"ms_per_step": round(wall1 * 1e3 / k1, 5), "Mpoints_s": round(N1 * k1 / wall1 / 1e6, 1), "steps": k1,
                         "latency_ms": {"p10": round(lat1[0], 4), "p50": round(lat1[1], 4), "p90": round(lat1[2], 4)},
                         "path": em1.last_update_path() if hasattr(l1, "emap_last_update_path") else "atomic"}
            em1.close()

    # ---- config.cfg4 / config.cfg5: BASELINE configs[3] / configs[4] on ONE GPU, timed in this process by the same code: the
    # denominators of every strong-scaling statement about those configurations (4096^2 / 4 M points with rays + overlap, `fp32`
    # indices; 8192^2 multi-modal map, 16 M points, RGB + 3 semantic layers).  Only next to the default workload at its own size.
    large = {}
    if a.workload == "cfg2" and not a.no_cfg3 and not a.no_large and not multimodal and C == 1024 and N == 1_000_000:
        import copy
        for name in ("cfg4", "cfg5"):
            b = copy.copy(a)
            b.workload, b.mode = name, "fp32"
            Cb, Nb = {"cfg4": (4096, 4_000_000), "cfg5": (8192, 16_000_000)}[name]
            mm = name == "cfg5"
            parb = parameter_from(workload_cfg(name), Cb, "fp32", weights); parb.device = local_rank
            emb = ElevationMap(parb)
            lb, cb = emb._lib, emb._ctx
            hostb = host_clouds(b, Cb, Nb, mm)[:2]
            devb = device_clouds(hip, hostb, True)
            del hostb
            specb = None
            if mm:
                specb = _lib.EmapSemSpec()
                specb.n_col, specb.col_chan[0], specb.col_layer[0] = 1, 3, 0
                specb.n_sum = 3
                for k_ in range(3):
                    specb.sum_chan[k_], specb.sum_layer[k_], specb.sum_kind[k_] = 4 + k_, 1 + k_, 0
                specb.alpha = 0.5
                if lb.emap_semantic_configure(cb, 4):
                    raise RuntimeError(lb.emap_last_error(cb).decode())

            def frb(i, stats=None, lb=lb, cb=cb, devb=devb, Nb=Nb, mm=mm, specb=specb, nz=1.0):
                rc = bind_cloud(lb, cb, devb[i % len(devb)], Nb)
                if mm:
                    rc = rc or mm_frame(lb, cb, Rp, tp, specb, stats, nz)
                else:
                    rc = rc or lb.emap_update(cb, Rp, tp, ct.c_double(1.0), ct.c_double(1.0), stats)
                if rc:
                    raise RuntimeError(lb.emap_last_error(cb).decode())
            warm(emb, frb)
            kb = 10
            wallb, msb, loopsb = timed(emb, frb, kb, loops=3)
            latb = latencies(emb, frb, 10)
            stb, visb = stage_profile(lb, cb, frb, 6, with_stats=not mm)
            sem_stage_ms = 0.0          # (fused inside the frame's tile pass: part of "fuse")
            if mm and not sem_in_frame(lb):       # the 11th stage: the RGB / semantic fusion (k_tile_semantic) -- a call of its own behind emap_update, timed by an event pair
                acc_s, e_ms = 0.0, ct.c_float(0)
                for i_ in range(6):
                    rc = bind_cloud(lb, cb, devb[i_ % len(devb)], Nb) or lb.emap_update(cb, Rp, tp, ct.c_double(1.0), ct.c_double(1.0), None)
                    lb.emap_timer_begin(cb)
                    rc = rc or lb.emap_semantic_update(cb, Rp, tp, ct.byref(specb))
                    lb.emap_timer_end(cb, ct.byref(e_ms))
                    if rc:
                        raise RuntimeError(lb.emap_last_error(cb).decode())
                    acc_s += e_ms.value
                sem_stage_ms = acc_s / 6
            gate_shut = None
            if mm:       # the same frames with the drift gate ruled out by the host (noise 0.0): no per-tile drift statistics
                import functools
                wall0, _, loops0 = timed(emb, functools.partial(frb, nz=0.0), kb, loops=3)
                gate_shut = {"ms_per_step": round(wall0 * 1e3 / kb, 5), "timed_loops_ms_per_step": loops0}
            Lb = Cb * Cb
            fbytes = 12 * Nb + 56 * Lb + (16 * Nb + 32 * Lb if mm else 0)
            rb = roofline(stb, ev_overhead, Nb, Lb, name, fbytes, msb / kb, visb, False)
            large[name] = {"workload": workload_text(b, Cb, Nb, mm), "index_mode": "fp32", "n_gpus": 1, "steps": kb,
                           "value": round(Nb * kb / wallb / 1e6, 2), "unit": "Mpoints/s", "ms_per_step": round(wallb * 1e3 / kb, 5),
                           "timed_loops_ms_per_step": loopsb, "latency_ms": {"p10": round(latb[0], 4), "p50": round(latb[1], 4), "p90": round(latb[2], 4)},
                           "dominant_kernel": rb["kernel"], "kernel_ms": rb["kernel_ms"], "bound": rb["bound"], "frac": rb["frac"], "hbm": rb.get("hbm"), "frame_frac": rb["frame_frac"],
                           "ray_visits_per_s": rb["ray_visits_per_s"], "stage_ms": (dict(rb["stage_ms"], semantic=round(sem_stage_ms, 5)) if mm else rb["stage_ms"]), "gate_shut": gate_shut,
                           "note": (("the RGB / semantic fusion is declared for the frame (emap_frame_semantics) and runs inside the 'fuse' stage's tile kernel on 32-byte sorted records; "
                                     if sem_in_frame(lb) else "stage_ms.semantic = the RGB / semantic fusion (k_tile_semantic: a call of its own behind the frame's ten stages, inside ms_per_step); ") +
                                    "cloud bound de-interleaved: (N, 3) xyz + (N, 4) channels, as emap_upload_points leaves an uploaded cloud") if mm else None}
            emb.close()
            free_clouds(hip, devb)

    # ---- the reference's real entry point: input_pointcloud with a HOST cloud (float64 as the ROS wrapper passes it, float32) -----
    # never part of `value`; the upload is asynchronous (host-side cast into pinned memory, DMA overlapping the previous frame)
    h2d = None
    if not multimodal:
        gbs = hip.pinned_h2d_gbs(24 * N)
        h2d = {"pinned_h2d_GBs": round(gbs, 1)}
        Rm = R.reshape(3, 3)
        for name, dt, bpp in (("f64", np.float64, 24), ("f32", np.float32, 12)):
            hosts = [np.ascontiguousarray(c[:, :3], dt) for c in clouds_host[:2]]
            for i in range(4):
                emap.input_pointcloud(hosts[i % 2], ["x", "y", "z"], Rm, t.copy(), 1.0, 1.0)
            emap.sync()
            k_in = max(4, min(a.steps, 30))
            t0 = time.perf_counter()
            for i in range(k_in):
                emap.input_pointcloud(hosts[i % 2], ["x", "y", "z"], Rm, t.copy(), 1.0, 1.0)
            emap.sync()
            dt_ms = (time.perf_counter() - t0) * 1e3 / k_in
            bound_ms = bpp * N / (gbs * 1e9) * 1e3
            h2d[name] = {"ms_per_frame": round(dt_ms, 4), "Mpoints_s": round(N / dt_ms / 1e3, 1), "pcie_bound_ms": round(bound_ms, 4),
                         "rate_vs_plain_pinned_copy": round(bound_ms / dt_ms, 3)}
        h2d["note"] = ("input_pointcloud(host cloud): frames per second through the reference's entry point; pcie_bound_ms = the time a plain "
                       "pinned hipMemcpy of the caller's bytes (24 / 12 per point) takes at pinned_h2d_GBs; rate_vs_plain_pinned_copy = "
                       "pcie_bound_ms / ms_per_frame (> 1 for float64 clouds: the upload casts on the host and moves 12 bytes per point, not 24)")

    if a.workload in ("cfg3", "cfg4"):
        config_cold = cold_start(emap, frame)
    else:
        config_cold = None
    cpu = None
    if not a.no_cpu_baseline and not multimodal:
        cpu = cpu_baseline(a, cfg, C, N, clouds_host, weights, R, t)

    config = {"workload": workload_text(a, C, N, multimodal), "index_mode": a.mode,
              "latency_ms": {"p10": round(p10, 4), "p50": round(p50, 4), "p90": round(p90, 4)},
              "device_ms_per_step": round(ms_dev / a.steps, 5), "cloud": "device resident (H2D excluded)",
              "timed_loops_ms_per_step": loops_ms, "value_from": "median of %d timed loops of %d frames" % (len(loops_ms), a.steps)}
    if config_cold:
        config["cold_start_ms"] = config_cold
    if cfg1:
        config["cfg1"] = cfg1
    if cfg3:
        config["cfg3"] = cfg3
    for name, rec in large.items():
        config[name] = rec
    if h2d:
        config["h2d_inclusive"] = h2d
    out = {
        "metric": "Mpoints/s fused (map-update p50 latency in config)", "value": round(mpts, 2), "unit": "Mpoints/s",
        "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": config, "roofline": roof, "cpu_baseline": cpu,
    }
    print(json.dumps(out), flush=True)


# -------------------------------------------------------------------------------------------------------------------------------
def _sem_spec():
    """RGB (packed 24 bit, colour fusion) + three averaged semantic channels: the multi-modal layers of BASELINE configs[4]"""
    from elevation_mapping_cupy_amd import _lib
    spec = _lib.EmapSemSpec()
    spec.n_col, spec.col_chan[0], spec.col_layer[0] = 1, 3, 0
    spec.n_sum = 3
    for k_ in range(3):
        spec.sum_chan[k_], spec.sum_layer[k_], spec.sum_kind[k_] = 4 + k_, 1 + k_, 0
    spec.alpha = 0.5
    return spec


def single_frame_ms(a, wl, hip, dev, C, N, frames=6, loops=3):
    """the SAME-BOX N = 1 denominator of a sharded sub-measurement: workload `wl` on one whole-map context of this device, median of
    `loops` timed loops of `frames` frames (ms per frame) -- run by rank 0 while the other ranks wait at a file barrier"""
    import copy
    from elevation_mapping_cupy_amd import _lib
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    from elevation_mapping_cupy_amd.configs import parameter_from
    b = copy.copy(a); b.workload = wl
    mode = "fp32" if C > 2049 else a.mode
    mm = wl == "cfg5"
    par = parameter_from(workload_cfg(wl), C, mode, load_weights()); par.device = dev
    em = ElevationMap(par); em.set_scatter_mode(a.scatter)
    lib, ctx = em._lib, em._ctx
    host = host_clouds(b, C, N, mm)[:2]
    devc = device_clouds(hip, host, True)
    del host
    spec = None
    if mm:
        spec = _sem_spec()
        if lib.emap_semantic_configure(ctx, 4):
            raise RuntimeError(lib.emap_last_error(ctx).decode())
    R = np.eye(3, dtype=np.float32).ravel().copy(); t = np.array([0, 0, 1], np.float32)
    Rp, tp = _lib.f32p(R), _lib.f32p(t)

    def fr(i):
        rc = bind_cloud(lib, ctx, devc[i % len(devc)], N)
        if mm:
            rc = rc or mm_frame(lib, ctx, Rp, tp, spec)
        else:
            rc = rc or lib.emap_update(ctx, Rp, tp, ct.c_double(1.0), ct.c_double(1.0), None)
        if rc:
            raise RuntimeError(lib.emap_last_error(ctx).decode())
    for i in range(3):
        fr(i)
        for _ in range(4):
            em.update_time()
    em.update_variance()
    for i in range(2):
        fr(i)
    res = []
    for _ in range(loops):
        em.sync(); t0 = time.perf_counter()
        for i in range(frames):
            fr(i)
        em.sync(); res.append((time.perf_counter() - t0) * 1e3 / frames)
    em.close(); free_clouds(hip, devc)
    return round(sorted(res)[len(res) // 2], 5)


def strips_workload(a, wl, C, N, steps, warmup, rank, world, dev, ndev, rdv, tag, hip):
    """Workload `wl` on `world` row strips, one per rank: a strip context, its own RCCL communicator (both exchange steps issued by the
    C library, emap_update_sharded), warm-up, a timed region of exactly `steps` frames (barrier + device sync on both sides, MAX over
    the ranks), per-frame latency and the per-stage device times of every rank -- the RGB / semantic fusion of a multi-modal frame
    INSIDE the timed frame and as an 11th stage ("semantic").  Returns None -- on every rank alike -- when the native communicator
    cannot be used, else a record (complete on rank 0)."""
    from elevation_mapping_cupy_amd import _lib, sharded
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    from elevation_mapping_cupy_amd.configs import parameter_from
    import copy
    b = copy.copy(a); b.workload = wl
    cfg = workload_cfg(wl)
    multimodal = wl == "cfg5"
    mode = "fp32" if C > 2049 else a.mode
    weights = load_weights()
    par = parameter_from(cfg, C, mode, weights, device=dev)
    halo = sharded.halo_rows_needed(par.dilation_size, world)
    rays = bool(cfg["enable_visibility_cleanup"])
    row_w = None
    # (a frame that marches its rays BY RAY wants equal heights: the library's predicate, not the map size alone)
    ray_mode = {"auto": 0, "by_row": 1, "by_ray": 2}.get(os.environ.get("EMAP_RAY_MODE", "auto"), 0)
    by_ray = rays and sharded.frame_marches_by_ray(C, N, world, "native", ray_mode, a.scatter)
    if rays and world > 1 and not by_ray and os.environ.get("EMAP_STRIPS", "balanced") == "balanced":
        row_w = sharded.ray_balanced_weights(C, float(cfg["resolution"]), float(cfg["max_ray_length"]), halo, world)
    r0, r1 = sharded.strip_rows(C, world, rank, row_w)
    ok, emap, err = True, None, ""
    try:
        if world > 1 and r1 - r0 < halo:
            raise ValueError("strip of %d rows is thinner than the %d-row halo" % (r1 - r0, halo))
        emap = ElevationMap(par, strip=(r0, r1 - r0, halo))
        emap.set_scatter_mode(a.scatter)
        if ray_mode:
            emap.set_ray_mode(os.environ["EMAP_RAY_MODE"])       # (A/B knob: "by_row" / "by_ray" on every rank)
    except Exception as ex:  # noqa: BLE001
        ok, err = False, str(ex)
    if not rdv.agree(tag + "create", ok):
        if err:
            print("[rank %d] strip context failed: %s" % (rank, err), file=sys.stderr)
        if emap is not None:
            emap.close()
        return None
    lib, ctx = emap._lib, emap._ctx
    path = sharded.rccl_library_path().encode()
    uid = (ct.c_uint8 * 128)()
    if rank == 0:
        ok = lib.emap_comm_unique_id(path, uid) == 0
        rdv.publish(tag + "uid", bytes(uid) if ok else b"")
    blob = rdv.fetch(tag + "uid", 0)
    if not rdv.agree(tag + "uid", len(blob) == 128):
        emap.close()
        return None
    uid = (ct.c_uint8 * 128).from_buffer_copy(blob)
    ok = lib.emap_comm_init(ctx, path, uid, rank, world) == 0
    if not ok:
        print("[rank %d] emap_comm_init: %s" % (rank, lib.emap_last_error(ctx).decode()), file=sys.stderr)
    if not rdv.agree(tag + "init", ok):
        emap.close()
        return None
    ok = lib.emap_comm_selftest(ctx) == 0
    if not rdv.agree(tag + "selftest", ok):
        lib.emap_comm_destroy(ctx); emap.close()
        return None

    nr = ct.c_int32(0)
    rccl_ranks = int(nr.value) if lib.emap_comm_count(ctx, ct.byref(nr)) == 0 and nr.value else None    # ncclCommCount of the live communicator

    def reduce(vals, op):
        buf = (ct.c_double * len(vals))(*vals)
        if lib.emap_comm_allreduce_host(ctx, buf, len(vals), op):
            raise RuntimeError(lib.emap_last_error(ctx).decode())
        return list(buf)

    def barrier():
        reduce([0.0], 0)

    clouds_host = host_clouds(b, C, N, multimodal)
    NCLOUD = len(clouds_host)
    channels = None
    if multimodal:
        channels = ["rgb", "sem0", "sem1", "sem2"]
        par.pointcloud_channel_fusions = {"rgb": "color", "default": "average"}
        emap.semantic_map.prepare(channels)
    R = np.eye(3, dtype=np.float32).ravel().copy()
    t = np.array([0, 0, 1], np.float32)
    Rp, tp = _lib.f32p(R), _lib.f32p(t)
    # Every rank is handed the same clouds; where the frame allows it (no visibility pass that marches by row) a rank keeps only the
    # points that can land in its rows -- the product's own predicate (emap_strip_point_mask = what emap_upload_points_strip /
    # ShardedElevationMap.input_pointcloud upload), applied once because the timed clouds are device resident.
    bucket = (world > 1 or os.environ.get("EMAP_BENCH_BUCKET") == "force") and (not rays or by_ray) and os.environ.get("EMAP_BENCH_BUCKET", "1") != "0"      # ("force": the bucketed code path on one rank, test hook)
    n_local = [N] * NCLOUD
    if bucket:
        local = []
        for p_ in clouds_host:
            q_ = np.ascontiguousarray(p_[emap.strip_point_mask(p_, R, t)])
            if q_.shape[0] == 0:                       # (no point of the cloud in this strip: one NaN row, as the upload path binds)
                q_ = np.full((1, p_.shape[1]), np.nan, np.float32)
            local.append(q_)
        n_local = [q_.shape[0] for q_ in local]
        clouds_dev = device_clouds(hip, local, not a.interleaved_cloud)
        del local
    else:
        clouds_dev = device_clouds(hip, clouds_host, not a.interleaved_cloud)
    sem_ms = ct.c_float(0)

    def frame(i, stats=None, time_sem=False):
        rc = bind_cloud(lib, ctx, clouds_dev[i % NCLOUD], n_local[i % NCLOUD])
        if bucket:
            rc = rc or lib.emap_declare_points_bucketed(ctx, Rp, tp, ct.c_int64(N))
        in_frame = multimodal and sem_in_frame(lib)
        if in_frame and not rc:         # the strip's RGB / semantic fusion rides inside the frame (emap_frame_semantics)
            emap.semantic_map.declare_frame(emap, channels)
        rc = rc or lib.emap_update_sharded(ctx, Rp, tp, ct.c_double(1.0), ct.c_double(1.0), stats)
        if rc:
            raise RuntimeError(lib.emap_last_error(ctx).decode())
        sem_ms.value = 0.0
        if multimodal and not in_frame:  # per strip, no exchange step: part of the frame (and of the timed region)
            if time_sem:
                lib.emap_timer_begin(ctx)
            emap.semantic_map.update_layers_pointcloud(emap, channels, R, t)
            if time_sem:
                lib.emap_timer_end(ctx, ct.byref(sem_ms))

    for i in range(3):
        frame(i)
        for _ in range(4):
            emap.update_time()
    emap.update_variance()
    for i in range(warmup):
        frame(i)
    emap.sync(); barrier()
    # ---- timed region: barrier + device sync on both sides, MAX over ranks ------------------------------------------------
    t0 = time.perf_counter()
    for i in range(steps):
        frame(i)
    emap.sync()
    wall_local = time.perf_counter() - t0
    barrier()
    wall = reduce([wall_local], 1)[0]
    # ---- per-frame latency: every frame synchronised on every rank (the collectives keep the ranks in step); max over ranks --------
    lat = []
    for i in range(min(steps, 40)):
        emap.sync()
        t1 = time.perf_counter(); frame(i); emap.sync(); lat.append((time.perf_counter() - t1) * 1e3)
    pct = reduce([float(x) for x in np.percentile(lat, [10, 50, 90])], 1)
    # ---- the same frames from a HOST cloud (what a sensor driver hands every rank): each rank's share bucketed, cast and uploaded by
    # emap_upload_points_strip inside the frame -- the per-rank host pass and PCIe that the device-resident timed region above leaves
    # out; asynchronous upload, so consecutive frames overlap it with the kernels.  Never part of `value`.
    host_frame_ms = None
    if bucket:
        def host_frame(i):
            p_ = clouds_host[i % NCLOUD]
            kept = ct.c_int64(0)
            rc = lib.emap_upload_points_strip(ctx, ct.c_void_p(p_.ctypes.data), ct.c_int64(p_.shape[0]), ct.c_int64(p_.shape[1]), 0, Rp, tp, ct.byref(kept))
            if multimodal and sem_in_frame(lib) and not rc:
                emap.semantic_map.declare_frame(emap, channels)
            rc = rc or lib.emap_update_sharded(ctx, Rp, tp, ct.c_double(1.0), ct.c_double(1.0), None)
            if rc:
                raise RuntimeError(lib.emap_last_error(ctx).decode())
            if multimodal and not sem_in_frame(lib):
                emap.semantic_map.update_layers_pointcloud(emap, channels, R, t)
        # (the upload alone first, no collective in it: a rank that cannot stage the cloud -- pinned-memory limits -- must not leave the
        # others waiting in the frame's all-reduce; every rank then agrees through the rendezvous before the frames are issued)
        p0_ = clouds_host[0]
        try:
            ok_up = lib.emap_upload_points_strip(ctx, ct.c_void_p(p0_.ctypes.data), ct.c_int64(p0_.shape[0]), ct.c_int64(p0_.shape[1]), 0, Rp, tp, ct.byref(ct.c_int64(0))) == 0
        except Exception:  # noqa: BLE001
            ok_up = False
        if rdv.agree(tag + "hostcloud", ok_up):
            for i in range(2):
                host_frame(i)
            emap.sync(); barrier()
            kh = max(3, min(steps, 6))
            t1 = time.perf_counter()
            for i in range(kh):
                host_frame(i)
            emap.sync()
            host_frame_ms = round(reduce([(time.perf_counter() - t1) * 1e3 / kh], 1)[0], 4)
            barrier()
        elif rank == 0:
            print("[bench] host-cloud frames skipped: a rank could not upload the cloud (%s)" % lib.emap_last_error(ctx).decode(), file=sys.stderr)
    # ---- per-stage device time of every rank's strip ------------------------------------------------------------------------------
    reps = min(steps, 20)
    stage_ms, _ = stage_profile(lib, ctx, frame, reps, with_stats=False)
    if multimodal:                      # the 11th stage: the RGB / semantic fusion of the strip (event pair around the call)
        acc = 0.0
        for i in range(reps):
            frame(i, None, True); acc += sem_ms.value
        stage_ms["semantic"] = acc / reps
    ev_overhead = event_overhead(lib, ctx)
    emap.sync(); barrier()
    all_stage = rdv.gather_json(tag + "stage_ms", {k: round(v, 5) for k, v in stage_ms.items()})
    rows = rdv.gather_json(tag + "rows", [int(r0), int(r1)])
    shares = rdv.gather_json(tag + "share", round(max(n_local) / float(N), 4))
    wb = ct.c_uint64(0)
    wire_bytes = int(wb.value) if lib.emap_comm_wire_bytes(ctx, ct.byref(wb)) == 0 else None
    wire_all = rdv.gather_json(tag + "wire", wire_bytes)       # sent + received per rank: the owners of the ray window's rows move the most
    rec = {"ok": True, "wall": wall, "stage_ms": stage_ms, "clouds_host": clouds_host, "cfg": cfg, "weights": weights, "R": R, "t": t}
    if rank == 0:
        L = C * C
        sb = sharded.strip_stage_bytes(N, L, world, full_sort=rays and not by_ray, bucketed=bucket)
        if multimodal:
            sb["semantic"] = (16 * N + 32 * L) / world        # 4 channels x 4 B per point, 4 layers x (read + write) per cell of the strip
        frame_bytes = 12 * N + 56 * L + (16 * N + 32 * L if multimodal else 0)
        roof = roofline({k: v for k, v in stage_ms.items() if k != "semantic"}, ev_overhead, N, L, wl, frame_bytes, wall * 1e3 / steps, 0, False,
                        stage_bytes={k: v for k, v in sb.items() if k != "semantic"})
        roof["rank"] = 0
        roof["per_rank_stage_ms"] = all_stage
        roof["note"] = ("rank 0's strip; 'gate' includes the all-reduce, 'post' the halo exchange overlapped with the interior stencils" +
                        ("; 'semantic' = the strip's RGB / semantic fusion (inside the timed frame)" if multimodal else ""))
        rec.update({
            "value": round(N * steps / wall / 1e6, 2), "ms_per_step": round(wall * 1e3 / steps, 5), "roofline": roof,
            "config": {"workload": workload_text(b, C, N, multimodal) + "; %d row strips, %s" % (world, "every rank binds only the points of its rows (bucketed with the library's predicate)" if bucket else "cloud replicated to every rank"),
                       "cloud_share_per_rank": shares,
                       "index_mode": mode, "latency_ms": {"p10": round(pct[0], 4), "p50": round(pct[1], 4), "p90": round(pct[2], 4)},
                       "halo_rows": halo, "parallelism": "row-strips x%d" % world, "ranks": world, "rccl_ranks": rccl_ranks,
                       "physical_devices": min(ndev, world),
                       "strip_rows": rows, "strip_heights": "equal ray work (thin around the sensor)" if row_w is not None else "equal",
                       "rays": ("by ray over an all-reduced window" if by_ray else "by row") if rays else "off",
                       "host_cloud_frame_ms": host_frame_ms, "by_ray_wire_bytes_per_frame": (max(x or 0 for x in wire_all) if by_ray else None), "by_ray_wire_bytes_per_rank": (wire_all if by_ray else None),
                       "collectives": "all-reduce(2 x f64) + neighbour halo send/recv per frame, RCCL issued by the C library "
                                      "(halo exchange in place on a second stream); bootstrap: file rendezvous, no torch",
                       "cloud": "device resident (H2D excluded)"}})
    rdv.barrier(tag + "measured")
    lib.emap_comm_destroy(ctx)
    emap.close()
    free_clouds(hip, clouds_dev)
    return rec


def strips_plan(a):
    """what `bench.py --gpus N` measures, in order: the workload asked for (the line's `value`: default cfg2 = BASELINE configs[1], the
    metric's configuration) and -- on the default line -- the two configurations north_star's scaling target names, as sharded
    sub-measurements with a same-box N = 1 run beside each.  EMAP_BENCH_SUB_SIZES="cfg5:1024:300000,cfg4:1024:300000" (test hook)
    shrinks the sub-measurements."""
    plan = [{"key": None, "tag": "main_", "workload": a.workload, "cell_n": a.cell_n, "points": a.points, "steps": a.steps, "warmup": a.warmup, "n1": False}]
    if a.workload == "cfg2" and not a.no_large and a.cell_n == 1024 and a.points == 1_000_000:
        sizes = {"cfg5": (8192, 16_000_000), "cfg4": (4096, 4_000_000)}
        for item in filter(None, os.environ.get("EMAP_BENCH_SUB_SIZES", "").split(",")):
            k_, c_, n_ = item.split(":"); sizes[k_] = (int(c_), int(n_))
        for wl in ("cfg5", "cfg4"):
            plan.append({"key": wl, "tag": wl + "_", "workload": wl, "cell_n": sizes[wl][0], "points": sizes[wl][1], "steps": min(a.steps, 10), "warmup": 2, "n1": True})
    return plan


def run_strips_native(a, rank, world, local_rank, rdv):
    """one row strip per rank, both exchange steps issued by the C library over RCCL; no torch in the process.
    Returns (False, None) -- on every rank alike -- when the native communicator cannot be used, so that the caller can fall
    back, else (True, the JSON object on rank 0).
    `value` / `ms_per_step` = the workload asked for (default cfg2 = BASELINE configs[1]: the metric's configuration).  The default
    line additionally carries what north_star's scaling target names -- `config.cfg5` (configs[4]: 8192^2 multi-modal map, 16 M points,
    semantic fusion inside the timed frame) and `config.cfg4` (configs[3]: 4096^2, 4 M points, rays + overlap) on the SAME N strips,
    each with the same-box N = 1 time beside it (`n1_ms_per_step`, `speedup_vs_n1`): the 1024^2 frame is seven launches of 5-20 us
    and does not strong-scale, the large maps are where the strips are for."""
    from elevation_mapping_cupy_amd import launch
    ndev = launch.device_count()
    if not rdv.agree("devices", ndev >= world or os.environ.get("EMAP_RCCL_LIB")):     # RCCL refuses two ranks on one device
        return False, None
    dev = local_rank % max(1, ndev)
    hip = Hip(); hip.set_device(dev)
    C, N = a.cell_n, a.points
    plan = strips_plan(a)
    main = strips_workload(a, a.workload, C, N, a.steps, a.warmup, rank, world, dev, ndev, rdv, plan[0]["tag"], hip)
    if main is None:
        return False, None
    subs = {}
    for item in plan[1:]:
            wl = item["key"]
            rec = strips_workload(a, wl, item["cell_n"], item["points"], item["steps"], item["warmup"], rank, world, dev, ndev, rdv, item["tag"], hip)
            n1 = None
            if rec is not None and rank == 0:
                try:
                    n1 = single_frame_ms(a, wl, hip, dev, item["cell_n"], item["points"])      # the same-box denominator (the other ranks wait below, off the GPU)
                except Exception as ex:  # noqa: BLE001
                    print("[rank 0] N = 1 run of %s failed: %s" % (wl, ex), file=sys.stderr)
            rdv.barrier(wl + "_n1")
            if rec is not None and rank == 0:
                sub = dict(rec["config"])
                sub.update({"value": rec["value"], "unit": "Mpoints/s", "ms_per_step": rec["ms_per_step"], "n_gpus": world,
                            "n1_ms_per_step": n1, "speedup_vs_n1": (round(n1 / rec["ms_per_step"], 3) if n1 else None),
                            "stage_ms_rank0": rec["roofline"]["stage_ms"], "per_rank_stage_ms": rec["roofline"]["per_rank_stage_ms"],
                            "dominant_kernel": rec["roofline"]["kernel"], "frac": rec["roofline"]["frac"]})
                subs[wl] = sub
    out = None
    if rank == 0:
        cpu = None
        if not a.no_cpu_baseline and a.workload != "cfg5":
            cpu = cpu_baseline(a, main["cfg"], C, N, main["clouds_host"], main["weights"], main["R"], main["t"])
        config = main["config"]
        config.update(subs)
        out = {
            "metric": "Mpoints/s fused (map-update p50 latency in config)", "value": main["value"],
            "unit": "Mpoints/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": main["ms_per_step"], "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": config, "roofline": main["roofline"], "cpu_baseline": cpu,
        }
        # what north_star's strong-scaling target is quoted on, where a reader of the line looks first: the 8192^2 multi-modal frame
        # (configs[4]) on these N strips against the same-box N = 1 frame.  `value` above stays BASELINE's metric configuration
        # (1024^2 / 1 M points), whose frame is seven launches of 5-20 us and does not strong-scale.
        c5 = subs.get("cfg5")
        if c5 and c5.get("speedup_vs_n1"):
            out["scaling_value"] = {"what": "config.cfg5 (8192^2 multi-modal map, 16 M points/frame) on %d row strips vs the same-box N = 1 frame" % world,
                                    "speedup_vs_n1": c5["speedup_vs_n1"], "ms_per_step": c5["ms_per_step"], "n1_ms_per_step": c5["n1_ms_per_step"],
                                    "rccl_ranks": c5.get("rccl_ranks"), "n_gpus": world,
                                    "excluded": "the per-rank host pass that buckets a HOST cloud by strip (emap_upload_points_strip) and PCIe: the timed clouds are device resident, bucketed once"}
    rdv.barrier("done")             # file barrier: the other ranks do not spin on the GPU while rank 0 runs the CPU baseline
    return True, out


def run_strips(a, rank, world, local_rank):
    from elevation_mapping_cupy_amd import launch
    rdv = launch.FileRendezvous.from_env(rank, world, timeout=900.0)
    if a.dry_run:                       # launcher / rendezvous plumbing only (CPU test hook)
        got = rdv.gather_json("dry", {"rank": rank, "pid": os.getpid()})
        ok = rdv.agree("dry", len(got) == world)
        plan = strips_plan(a)
        for item in plan:               # the rendezvous steps of every planned measurement, under the tags the real run uses
            for step in ("create", "uid", "init", "selftest"):
                ok = rdv.agree(item["tag"] + step, True) and ok
            rdv.barrier(item["tag"] + "measured")
            if item["n1"]:
                rdv.barrier(item["key"] + "_n1")
        rdv.barrier("dry_done")
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks": got, "agreed": ok,
                              "plan": {"value": {k: plan[0][k] for k in ("workload", "cell_n", "points", "steps")},
                                       "config": {it["key"]: {k: it[k] for k in ("workload", "cell_n", "points", "steps", "n1")} for it in plan[1:]}}}), flush=True)
        rdv.finish()
        return
    # RCCL prints its version banner on stdout through C stdio: keep fd 1 pointed at stderr until the JSON line is due
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    status, out = False, None
    try:
        if os.environ.get("EMAP_COMM", "native") == "native":
            status, out = run_strips_native(a, rank, world, local_rank, rdv)
    finally:
        try:
            ct.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if status:
        if rank == 0:
            print(json.dumps(out), flush=True)
    else:
        if rank == 0:
            print("native RCCL strips unavailable here (fewer devices than ranks, or RCCL failed): torch.distributed fallback", file=sys.stderr)
        sys.path.insert(0, os.path.join(ROOT, "tests"))      # (test infrastructure: the product package holds no torch.distributed path)
        import _torch_strips
        _torch_strips.bench_main(a, rank, world, local_rank)
    rdv.finish()


# -------------------------------------------------------------------------------------------------------------------------------
SHIPPED_PLUGINS = """
min_filter: {enable: True, fill_nan: False, is_height_layer: True, layer_name: "min_filter", extra_params: {dilation_size: 1, iteration_n: 30}}
smooth_filter: {enable: True, fill_nan: False, is_height_layer: True, layer_name: "smooth", extra_params: {input_layer_name: "min_filter"}}
inpainting: {enable: True, fill_nan: False, is_height_layer: True, layer_name: "inpaint", extra_params: {method: "telea"}}
erosion: {enable: True, fill_nan: False, is_height_layer: False, layer_name: "erosion", extra_params: {input_layer_name: "traversability", dilation_size: 3, iteration_n: 20, reverse: True}}
"""


def _pct(xs):
    return {"p10": round(float(np.percentile(xs, 10)) * 1e3, 4), "p50": round(float(np.percentile(xs, 50)) * 1e3, 4), "p90": round(float(np.percentile(xs, 90)) * 1e3, 4)}


def run_ref_main(a, local_rank=0):
    """The boundary, not just the ABI: the loop of the reference's own profiling script (EM/elevation_mapping.py:925-967 -- 50 x
    {input_pointcloud of a 100 000 x 7 HOST float64 cloud, update_normal, move_to, seven get_map_with_name_ref, one
    get_polygon_traversability}) through the drop-in package (compat/elevation_mapping_cupy), every call timed on the host; and one
    stage-timed line each for the rows of SURVEY section 8 that had no number: the MinFilter plugin (a15, as shipped: 1 x 30 sweeps), the
    layer read-back (f3: emap_publish_layer) and the camera path (f4: emap_image_correspondence + emap_image_fuse)."""
    import pickle
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "compat"))
    from elevation_mapping_cupy import elevation_mapping as em_mod
    from elevation_mapping_cupy import parameter as par_mod
    from elevation_mapping_cupy_amd import _lib
    d = tempfile.mkdtemp(prefix="emap_refmain_")
    w = load_weights()
    with open(os.path.join(d, "weights.dat"), "wb") as f:
        pickle.dump({"conv1.weight": w["w1"], "conv2.weight": w["w2"], "conv3.weight": w["w3"], "conv_final.weight": w["w_out"]}, f)
    with open(os.path.join(d, "plugin_config.yaml"), "w") as f:
        f.write(SHIPPED_PLUGINS)
    param = par_mod.Parameter(use_chainer=False, weight_file=os.path.join(d, "weights.dat"), plugin_config_file=os.path.join(d, "plugin_config.yaml"))
    param.additional_layers = ["rgb", "grass", "tree", "people"]
    # (the script's `param.fusion_algorithms = ["color", "class_bayesian" x 3]` is the pre-plugin spelling: in the reference's current
    # API that list names the fusion PLUGINS to register, so the intent -- rgb by colour, the classes by class_bayesian -- goes here)
    param.pointcloud_channel_fusions = {"rgb": "color", "default": "class_bayesian"}
    param.update()
    param.device = local_rank
    elevation = em_mod.ElevationMap(param)
    lib, ctx = elevation._lib, elevation._ctx
    layers = ["elevation", "variance", "traversability", "min_filter", "smooth", "inpaint", "rgb"]
    rng = np.random.default_rng(123)
    R, t = rng.random((3, 3)), rng.random(3)
    points = rng.random((100000, len(layers)))                      # float64, as xp.random.rand gives it
    channels = ["x", "y", "z"] + list(param.additional_layers)
    data = np.zeros((elevation.cell_n - 2, elevation.cell_n - 2), dtype=np.float32)
    calls = {k: [] for k in ["input_pointcloud", "update_normal", "move_to", "get_map_with_name_ref x7", "get_polygon_traversability", "iteration"]}
    per_layer = {k: [] for k in layers}

    def timed_call(key, fn):
        t0 = time.perf_counter(); r = fn(); calls[key].append(time.perf_counter() - t0); return r

    def get_all():
        for layer in layers:
            t0 = time.perf_counter(); elevation.get_map_with_name_ref(layer, data); per_layer[layer].append(time.perf_counter() - t0)
    n_it = max(10, a.steps)
    for i in range(a.warmup + n_it):
        if i == a.warmup:
            for v in list(calls.values()) + list(per_layer.values()):
                del v[:]
        t_it = time.perf_counter()
        timed_call("input_pointcloud", lambda: elevation.input_pointcloud(points, channels, R, t.copy(), 0, 0))
        timed_call("update_normal", lambda: elevation.update_normal(elevation.elevation_map[0]))
        pos = np.array([i * 0.01, i * 0.02, i * 0.01])
        timed_call("move_to", lambda: elevation.move_to(pos, R))
        timed_call("get_map_with_name_ref x7", get_all)
        polygon = np.array([[0, 0], [2, 0], [0, 2]], dtype=np.float64)
        result = np.array([0, 0, 0], np.float64)
        timed_call("get_polygon_traversability", lambda: elevation.get_polygon_traversability(polygon, result))
        calls["iteration"].append(time.perf_counter() - t_it)
    elevation.sync() if hasattr(elevation, "sync") else None
    # device time of the frame inside input_pointcloud (the stage events of emap_update) next to the call's host time
    lib.emap_enable_stage_timing(ctx, 1)
    dev_ms = []
    for i in range(10):
        elevation.input_pointcloud(points, channels, R, t.copy(), 0, 0)
        ms10 = (ct.c_float * 10)(); lib.emap_get_stage_times(ctx, ms10); dev_ms.append(float(sum(ms10)))
    lib.emap_enable_stage_timing(ctx, 0)
    frame_dev_ms = float(np.median(dev_ms))
    wall = float(np.sum(calls["iteration"]))
    it_ms = float(np.median(calls["iteration"])) * 1e3

    # ---- stage-timed lines: device time by an event pair on the context's stream around the call, host time beside it ---------------
    def staged(fn, reps=20):
        host, dev = [], []
        e_ms = ct.c_float(0)
        for _ in range(reps + 2):
            lib.emap_sync(ctx)
            lib.emap_timer_begin(ctx); t0 = time.perf_counter(); rc = fn(); th = time.perf_counter() - t0; lib.emap_timer_end(ctx, ct.byref(e_ms))
            if rc:
                raise RuntimeError(lib.emap_last_error(ctx).decode())
            host.append(th); dev.append(e_ms.value)
        return round(float(np.median(host[2:])) * 1e3, 4), round(float(np.median(dev[2:])), 4)

    stages = {}
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap as AmdMap
    from elevation_mapping_cupy_amd.configs import parameter_from
    for C in (202, 1024):
        par = parameter_from(workload_cfg("cfg2"), C, "reference_fp16", w); par.device = local_rank
        m = AmdMap(par)
        ml, mc = m._lib, m._ctx
        import _fixtures as fx
        Rm, tm = fx.POSES["identity"]
        for f_ in range(3):
            m.update_map_with_kernel(fx.cloud(C, 50000 if C == 202 else 1_000_000, f_), [], Rm, tm.copy(), 1.0, 1.0)
        L = C * C
        out = np.empty((C, C), np.float32); sweeps = ct.c_int32(0)
        lib_, ctx_ = lib, ctx
        lib, ctx = ml, mc                       # (staged() times on the context it is given)
        h_ms, d_ms = staged(lambda: ml.emap_min_filter(mc, None, None, 1, 30, _lib.f32p(out), ct.byref(sweeps)))
        nsw = max(1, int(sweeps.value))
        stages["min_filter_%d" % C] = {"what": "emap_min_filter (a15, MinFilter plugin as shipped: dilation 1, <= 30 Jacobi sweeps with a device-side early exit), the map's own planes, result read back",
                                        "cells": L, "sweeps_run": nsw, "host_ms": h_ms, "device_ms": d_ms, "Gcells_per_s": round(L * nsw / (d_ms * 1e-3) / 1e9, 2),
                                        "algorithmic_bytes": int(16 * L * nsw + 4 * L), "frac_hbm": round((16.0 * L * nsw + 4 * L) / (d_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        pub = np.empty((C - 2, C - 2), np.float32)
        h_ms, d_ms = staged(lambda: ml.emap_publish_layer(mc, 0, ct.c_float(0.0), 0, _lib.f32p(pub)))
        stages["publish_layer_%d" % C] = {"what": "emap_publish_layer (f3: get_map_with_name_ref('elevation') -- strip, NaN fill, + center_z, double flip in one kernel + one D2H of (C-2)^2 floats)",
                                           "cells": L, "host_ms": h_ms, "device_ms": d_ms, "Gcells_per_s": round(L / (d_ms * 1e-3) / 1e9, 2),
                                           "algorithmic_bytes": int(8 * L + 4 * (C - 2) ** 2), "d2h_GBs": round(4.0 * (C - 2) ** 2 / (h_ms * 1e-3) / 1e9, 2),
                                           "frac_hbm": round((8.0 * L + 4 * (C - 2) ** 2) / (d_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        # camera path: a 480 x 640 pinhole looking down at the map from 3 m, one feature plane (exponential fusion needs a layer)
        if ml.emap_semantic_configure(mc, 1):
            raise RuntimeError(ml.emap_last_error(mc).decode())
        H, W = 480, 640
        K = np.array([[400, 0, W / 2], [0, 400, H / 2], [0, 0, 1]], np.float32)
        Rc = np.array([[1, 0, 0], [0, -1, 0], [0, 0, -1]], np.float32); tc = np.array([0, 0, 3], np.float32)
        Pm = (K @ np.concatenate([Rc, tc[:, None]], 1)).astype(np.float32)
        D = np.zeros(5, np.float32); cen = np.zeros(3, np.float32)
        img = np.random.default_rng(3).random((1, H, W), dtype=np.float32)
        h1, d1 = staged(lambda: ml.emap_image_correspondence(mc, ct.c_float(C / 2), ct.c_float(C / 2), ct.c_float(3.0), _lib.f32p(np.ascontiguousarray(Pm.reshape(-1))),
                                                             _lib.f32p(np.ascontiguousarray(K.reshape(-1))), _lib.f32p(D), ct.c_float(H), ct.c_float(W), _lib.f32p(cen)))
        h2, d2 = staged(lambda: ml.emap_image_fuse(mc, 0, 0, _lib.f32p(img), 1, H, W, ct.c_double(0.7)))
        stages["image_%d" % C] = {"what": "f4 camera path: emap_image_correspondence (projection + Bresenham occlusion walk per cell) then emap_image_fuse (one 480 x 640 plane uploaded, exponential fusion)",
                                   "cells": L, "correspondence": {"host_ms": h1, "device_ms": d1, "Gcells_per_s": round(L / (d1 * 1e-3) / 1e9, 2)},
                                   "fuse": {"host_ms": h2, "device_ms": d2, "Gcells_per_s": round(L / (d2 * 1e-3) / 1e9, 2), "image_bytes": int(img.nbytes)}}
        lib, ctx = lib_, ctx_
        m.close()
    out = {"metric": "reference profiling loop (EM/elevation_mapping.py:925-967) through the drop-in package, iterations/s", "value": round(n_it / wall, 2),
           "unit": "iterations/s", "n_gpus": 1, "steps": n_it, "warmup": a.warmup, "ms_per_step": round(it_ms, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "source_stamp": source_stamp(),
           "config": {"workload": "ref_main: %dx%d map (Parameter defaults), 100000 x 7 host float64 cloud per iteration (rgb: color, grass / tree / people: class_bayesian), "
                                  "shipped plugin configuration (min_filter 1 x 30, smooth, inpaint telea, erosion), seven layers read back per iteration" % (elevation.cell_n, elevation.cell_n),
                      "host_ms_per_call": {k: _pct(v) for k, v in calls.items()},
                      "get_map_with_name_ref_ms": {k: _pct(v) for k, v in per_layer.items()},
                      "input_pointcloud": {"host_ms_p50": _pct(calls["input_pointcloud"])["p50"], "frame_device_ms": round(frame_dev_ms, 4),
                                           "note": "device = sum of the frame's stage event spacings; the call returns without waiting for the device (no statistics read back)"},
                      "stage_lines": stages}}
    print(json.dumps(out), flush=True)


def launch_ranks(a, argv):
    """bench.py was started without a launcher: become one (one process per GPU, WORLD_SIZE = --gpus)"""
    from elevation_mapping_cupy_amd import launch
    rc, out0 = launch.spawn_ranks(os.path.abspath(__file__), argv, a.gpus, timeout=3000)
    lines = [l for l in out0.splitlines() if l.strip().startswith("{")]
    if rc != 0 or not lines:
        sys.stderr.write(out0)
        raise SystemExit("bench.py --gpus %d: ranks failed (rc %d)" % (a.gpus, rc))
    print(lines[-1], flush=True)


def main():
    argv = sys.argv[1:]
    a = parse(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(a, argv)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or a.force_sharded or a.dry_run:
        return run_strips(a, rank, world, local_rank)
    if a.workload == "ref_main":
        return run_ref_main(a, local_rank)
    return run_single(a, local_rank)


if __name__ == "__main__":
    main()
