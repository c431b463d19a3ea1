"""GPU: the library's own RCCL communicator (emap_comm_init / emap_update_sharded).  One MI355X is available to the tests, so
the communicator is exercised with a single rank: RCCL is dlopen()ed, a real ncclCommInitRank / ncclAllReduce / grouped
ncclSend + ncclRecv run on the device (emap_comm_selftest), and a frame through emap_update_sharded must equal emap_update
bit for bit.  The multi-rank stage order is the one tests/test_hip_strips.py and tests/test_sharded_gloo.py verify."""
import numpy as np
import pytest

import _fixtures as fx

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg_name,C,N", [("yaml", 130, 40000), ("yaml_norays", 1024, 300000)])
def test_native_comm_single_rank_frame_equals_emap_update(cfg_name, C, N, weights):
    import torch
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    from elevation_mapping_cupy_amd.sharded import HipStripEngine, NativeComm, ShardedElevationMap, rccl_library_path
    from oracle import emap_oracle as eo
    cfg = dict(eo.DEFAULTS); cfg.update(eo.YAML)
    if cfg_name.endswith("norays"):
        cfg["enable_visibility_cleanup"] = False
    R, t = fx.POSES["rotated"]
    clouds = [fx.cloud(C, N, f, dz=dz) for f, dz in enumerate((0.0, -0.02, -0.1))]
    full = ElevationMap(parameter_from(cfg, C, "reference_fp16", weights))
    for p in clouds:
        full.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0)
        for _ in range(6):
            full.update_time()
    eng = HipStripEngine(parameter_from(cfg, C, "reference_fp16", weights), 0, 1, 0, torch.device("cuda", 0))
    comm = NativeComm(eng, rank=0, world=1, bootstrap=False)
    assert "rccl" in rccl_library_path()
    comm.selftest()
    sm = ShardedElevationMap(eng, comm, cfg["enable_visibility_cleanup"], cfg["enable_overlap_clearance"])
    for p in clouds:
        eng.bind_points(p)
        sm.update(R, t, 1.0, 1.0)
        for _ in range(6):
            eng.update_time()
    eng.sync()
    assert eng.map.elevation_map.tobytes() == full.elevation_map.tobytes()
    assert eng.map.normal_map.tobytes() == full.normal_map.tobytes()
    assert eng.map.get_additive_mean_error() == full.get_additive_mean_error()
    comm.selftest()       # still healthy after the frames
    assert eng.lib.emap_comm_destroy(eng.ctx) == 0


def test_update_sharded_without_communicator_fails_loudly(weights):
    from elevation_mapping_cupy_amd._lib import EmapError
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    from oracle import emap_oracle as eo
    import ctypes as ct
    m = ElevationMap(parameter_from(dict(eo.DEFAULTS), 66, "reference_fp16", weights))
    m.bind_points(fx.cloud(66, 1000, 0))
    R = np.eye(3, dtype=np.float32).ravel().copy(); t = np.array([0, 0, 1], np.float32)
    rc = m._lib.emap_update_sharded(m._ctx, R.ctypes.data_as(ct.POINTER(ct.c_float)), t.ctypes.data_as(ct.POINTER(ct.c_float)),
                                    ct.c_double(0), ct.c_double(0), None)
    assert rc != 0
    with pytest.raises(EmapError, match="emap_comm_init"):
        m._chk(rc)
    bad = (ct.c_uint8 * 128)()
    assert m._lib.emap_comm_init(m._ctx, b"/nonexistent/librccl.so", bad, 0, 1) != 0
    assert b"dlopen" in m._lib.emap_last_error(m._ctx)


def _fake_rccl(kind="blocking"):
    """in-process stand-in for RCCL (ranks are threads on one GPU): tests/fake_rccl/, built by tests/_util.py"""
    from _util import rccl_stand_in
    return rccl_stand_in(kind)


def _strips_vs_single(world, cfg, C, frames, scatter, weights, mode="reference_fp16", check_gather=None, stand_in="blocking", ray_mode="auto"):
    """frames = [(cloud, R, t, position_noise, orientation_noise, n_update_time, move_to vector or None), ...]: one single-context
    map and `world` strip contexts (threads, in-process RCCL stand-in) run the same frames; every strip must equal the rows of the
    single-context map bit for bit.  Returns nothing; asserts."""
    import ctypes as ct
    import threading
    import torch
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    from elevation_mapping_cupy_amd.sharded import HipStripEngine, NativeComm, ShardedElevationMap
    lib_path = _fake_rccl(stand_in)
    full = ElevationMap(parameter_from(cfg, C, mode, weights))
    full.set_scatter_mode(scatter)
    t_rel = []                                                    # the map-centre relative translation the single context computes (float32: t_world - center)
    for p, R, t, pn, on, nt, mv in frames:
        tw = (t + full.center).astype(np.float32)
        t_rel.append(tw - full.center)
        full.update_map_with_kernel(p, [], R, tw, pn, on)
        for _ in range(nt):
            full.update_time()
        if mv is not None:
            full.move_to(np.array(mv, np.float64), np.eye(3))
    want, want_n, want_add = full.elevation_map, full.normal_map, full.get_additive_mean_error()
    # one id for all ranks (what the bootstrap channel distributes in a real launch)
    uid = (ct.c_uint8 * 128)()
    assert full._lib.emap_comm_unique_id(lib_path.encode(), uid) == 0
    dev = torch.device("cuda", 0)
    out, errs, wires = [None] * world, [], [0] * world

    def run(rank):
        try:
            eng = HipStripEngine(parameter_from(cfg, C, mode, weights), rank, world, 0, dev)
            eng.map.set_scatter_mode(scatter)
            eng.map.set_ray_mode(ray_mode)
            comm = NativeComm(eng, rank=rank, world=world, bootstrap=False, uid=bytes(uid), rccl_path=lib_path)
            comm.selftest()
            assert comm.rccl_ranks() == world
            sm = ShardedElevationMap(eng, comm, cfg["enable_visibility_cleanup"], cfg["enable_overlap_clearance"])
            for (p, R, _t, pn, on, nt, mv), t in zip(frames, t_rel):
                eng.bind_points(p)
                sm.update(R, t, pn, on)                          # (map-centre relative t: the sensor rides with the centre)
                for _ in range(nt):
                    eng.update_time()
                if mv is not None:
                    sm.move_to(np.array(mv, np.float64), np.eye(3))
            eng.sync()
            wb = ct.c_uint64(0)
            assert eng.lib.emap_comm_wire_bytes(eng.ctx, ct.byref(wb)) == 0
            wires[rank] = int(wb.value)
            gathered = (sm.gather("elevation"), sm.gather("normal_z")) if check_gather else None      # collective read-back of whole planes
            out[rank] = (eng.map.logical_row_begin, eng.map.rows, eng.map.elevation_map, eng.map.normal_map, eng.map.get_additive_mean_error(), gathered)
            eng.lib.emap_comm_destroy(eng.ctx)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    [x.start() for x in th]
    [x.join(timeout=120) for x in th]
    assert not any(x.is_alive() for x in th), "a rank is stuck in the exchange"
    assert not errs, errs
    import os
    if ray_mode == "by_ray" and world >= 3 and cfg["enable_visibility_cleanup"] and any(wires) and os.environ.get("EMAP_BYRAY_ALLREDUCE", "0") == "0":
        # rays by ray: the window is BROADCAST by the owners of its rows and the effects are REDUCED to them (32 + 20 bytes per window cell of
        # the map, each moved W - 1 times, counted once by the sender and once by the receiver) -- not three all-reduces, where every rank
        # would report the same payload
        assert sum(wires) % (2 * (world - 1) * 52) == 0, wires
        if float(cfg["max_ray_length"]) < 5.0:      # a real sub-window: two or three ranks own its rows and move more than the others
            assert len(set(wires)) > 1, wires
    for b, rows, m, nm, add, gathered in out:
        if gathered is not None and check_gather == "all":        # (after a move the gathered normal planes are compared only once a frame has rewritten them)
            assert np.array_equal(gathered[0], want[0]) and np.array_equal(gathered[1], want_n[2]), "gathered planes differ"
        elif gathered is not None:
            assert np.array_equal(gathered[0], want[0]), "gathered elevation differs"
        idx = (b + np.arange(rows)) % C                           # the strip's view: logical rows b, b + 1, ... of the full map
        assert m.tobytes() == np.take(want, idx, axis=1).tobytes(), "strip whose view starts at logical row %d differs" % b
        assert nm.tobytes() == np.take(want_n, idx, axis=1).tobytes(), "normals of the strip at logical row %d differ" % b
        assert add == want_add


@pytest.mark.parametrize("stand_in", ["blocking", "stream"])
@pytest.mark.parametrize("moves", [False, True])
@pytest.mark.parametrize("world,cfg_name,C,N,scatter", [(2, "yaml", 130, 40000, "auto"), (3, "yaml_norays", 202, 60000, "auto"), (4, "default", 202, 40000, "auto"),
                                                        # the tile-binned scatter on strips: without a visibility pass the point passes run the cheap
                                                        # ownership test + lane compaction (k_bin_hist / k_bin_scatter<.., STRIP>), with one they keep the
                                                        # ray-only bin; 8 strips of the 1024^2 map with a cloud large enough for the automatic choice
                                                        (3, "yaml_norays", 202, 60000, "binned"), (4, "yaml", 130, 40000, "binned"),
                                                        (8, "yaml_norays", 1024, 300000, "auto")])
def test_native_multi_rank_frame_with_in_process_rccl_stand_in(world, cfg_name, C, N, scatter, moves, stand_in, weights):
    """emap_comm_init + emap_update_sharded with SEVERAL ranks: the library's own orchestration (all-reduce between count and fuse,
    in-place halo send / recv on the second stream, interior / boundary stencil split) driven through a stand-in for the nine RCCL
    entry points whose ranks are threads of this process (RCCL refuses two ranks on one GPU) -- two independent stand-ins: one that
    synchronises with the host around every copy, one whose collectives are stream ordered like RCCL's (events across the ranks'
    streams, tests/fake_rccl/stream_rccl.hip).  Every strip must equal the rows of the single-context map bit for bit."""
    from oracle import emap_oracle as eo
    cfg = dict(eo.DEFAULTS); cfg.update(eo.YAML if cfg_name.startswith("yaml") else {})
    if cfg_name.endswith("norays"):
        cfg["enable_visibility_cleanup"] = False
    R, t = fx.POSES["rotated"]
    # move_to between the frames: ring halo, normal rows fetched from their owners
    MV = [(0.13, -0.3, 0.05), (-0.10, 0.17, -0.02), None] if moves else [None] * 3
    frames = [(fx.cloud(C, N, f, dz=dz), R, t, 1.0, 1.0, 6, mv) for (f, dz), mv in zip(enumerate((0.0, -0.02, -0.1)), MV)]
    _strips_vs_single(world, cfg, C, frames, scatter, weights, check_gather=None if scatter != "binned" else ("elevation" if moves else "all"), stand_in=stand_in)


@pytest.mark.parametrize("k", range(10))
def test_fuzz_strips_bitwise(k, weights):
    """seeded random strip scenarios (world size, map size, index mode, feature toggles, scatter path, per-frame poses, decay passes,
    small map moves between frames): strips == single context, bit for bit"""
    from oracle import emap_oracle as eo
    rng = np.random.default_rng(9100 + k)
    cfg = dict(eo.DEFAULTS); cfg.update(eo.YAML if rng.random() < 0.6 else {})
    for key in ("enable_visibility_cleanup", "enable_overlap_clearance", "enable_edge_sharpen", "enable_drift_compensation"):
        cfg[key] = bool(rng.random() < 0.7)
    cfg["dilation_size"] = int(rng.integers(1, 4))
    C = int(rng.choice([130, 157, 202, 257, 300]))
    world = int(rng.integers(2, 7))
    mode = "reference_fp16" if rng.random() < 0.6 else "fp32"
    scatter = ["auto", "binned", "atomic"][int(rng.integers(0, 3))]
    frames = []
    for f in range(3):
        a = rng.uniform(-0.4, 0.4, 3)
        R = fx.rot(*a)
        t = np.array([rng.uniform(-0.6, 0.6), rng.uniform(-0.6, 0.6), rng.uniform(0.7, 1.3)], np.float32)
        p = fx.cloud(C, int(rng.integers(5000, 60000)), 50 * k + f, dz=float(rng.uniform(-0.2, 0.05)))
        if rng.random() < 0.5:
            p[::211, int(rng.integers(0, 3))] = np.nan
        noise = 1.0 if rng.random() < 0.6 else 0.0
        mv = (float(rng.uniform(-0.2, 0.2)), float(rng.uniform(-0.3, 0.3)), 0.0) if (f < 2 and rng.random() < 0.5) else None     # (small moves here; tens of rows: test_strips_after_large_map_moves)
        frames.append((p, R, t, noise, noise, int(rng.integers(0, 9)), mv))
    _strips_vs_single(world, cfg, C, frames, scatter, weights, mode=mode, stand_in="stream" if k % 2 else "blocking",
                      ray_mode=["auto", "by_ray", "by_ray"][int(rng.integers(0, 3))])          # (by ray takes effect on the tile-binned frames with a visibility pass)


@pytest.mark.parametrize("stand_in", ["blocking", "stream"])
@pytest.mark.parametrize("world,cfg_name,C,N,moves", [(2, "yaml", 130, 40000, False), (3, "yaml", 202, 60000, True), (4, "default", 300, 60000, True),
                                                      (8, "default", 300, 90000, False), (4, "default_fp32", 421, 70000, True)])
def test_rays_by_ray_reproduce_the_single_context(world, cfg_name, C, N, moves, stand_in, weights):
    """The visibility pass of a sharded frame BY RAY (emap_set_ray_mode 2): every rank marches the rays of the points of its rows over
    the all-reduced window around the sensor, the effects are all-reduced back to the owners of the rows -- the same visits as on
    one GPU, so every strip must equal the rows of the single-context map bit for bit.  `yaml`: max_ray_length 10 m, the window is
    the whole map; `default`: 2 m, a real sub-window (about 116 x 128 cells of a 300^2 map) whose position follows the sensor;
    with map moves the circular origin, the normals' own origin and the ring of strips are exercised as well."""
    from oracle import emap_oracle as eo
    cfg = dict(eo.DEFAULTS); cfg.update(eo.YAML if cfg_name.startswith("yaml") else {})
    cfg["enable_visibility_cleanup"] = True
    mode = "fp32" if cfg_name.endswith("fp32") else "reference_fp16"
    R, t = fx.POSES["rotated"]
    MV = [(0.13, -0.3, 0.05), (-0.10, 0.17, -0.02), None] if moves else [None] * 3
    frames = [(fx.cloud(C, N, f, dz=dz), R, t + np.array([0.4 * f, -0.3 * f, 0], np.float32), 1.0, 1.0, 6, mv) for (f, dz), mv in zip(enumerate((0.0, -0.02, -0.1)), MV)]
    _strips_vs_single(world, cfg, C, frames, "binned", weights, mode=mode, stand_in=stand_in, ray_mode="by_ray")


@pytest.mark.parametrize("world,ray_mode,stand_in", [(2, "by_row", "blocking"), (2, "by_ray", "stream"), (4, "by_row", "stream"), (4, "by_ray", "blocking"),
                                                      (8, "by_row", "blocking"), (8, "by_ray", "stream"), (3, "by_ray", "blocking"), (5, "by_row", "stream")])
def test_strips_after_large_map_moves(world, ray_mode, stand_in, weights):
    """A robot that moves FAST: move_to of 20-45 rows between the frames (0.8-1.8 m at 10 Hz), more than halo_rows (7) and -- on 8
    ranks of a 300-row map -- more than a strip is high (37 rows).  The normal planes are not shifted with the map (reference
    elevation_mapping.py:200-214: normal_map stays), so after the move the visibility pass of the next frame reads, for every cell, the
    STALE normal at the cell's old index (custom_kernels.py:243-246) -- rows that a strip does not hold any more.  Until round 4 a
    strip read zeros there (a documented deviation); now every rank fetches the rows its cells belong to from whoever owns them
    (emap_api.hip: normal_exchange): strips == single context, bit for bit, by row and by ray."""
    from oracle import emap_oracle as eo
    cfg = dict(eo.DEFAULTS); cfg.update(eo.YAML)
    cfg["enable_visibility_cleanup"] = True
    C, N = 300, 60000
    R, t = fx.POSES["rotated"]
    MV = [(1.2, -0.5, 0.0), (-0.6, 0.3, 0.02), (0.8, 1.1, 0.0), None]           # cumulative targets: +30, -45, +35 rows (and column shifts)
    frames = [(fx.cloud(C, N, 30 + f, dz=dz), R, t + np.array([0.2 * f, -0.1 * f, 0], np.float32), 1.0, 1.0, 6, mv)
              for (f, dz), mv in zip(enumerate((0.0, -0.02, -0.06, -0.1)), MV)]
    _strips_vs_single(world, cfg, C, frames, "binned", weights, stand_in=stand_in, ray_mode=ray_mode)
