"""GPU: the strip kernels (row0/nrows/halo addressing, owned-row filter of points and rays, device-side drift sums,
halo pack/unpack) -- G strip contexts on ONE device driven in lock-step by threads must reproduce the single-context
map bit for bit.  (The multi-process exchange itself is covered on CPU by tests/test_sharded_gloo.py; the driver's
multi-GPU run is emap_update_sharded over RCCL: test_hip_comm.py.)"""
import threading

import numpy as np
import pytest

import _fixtures as fx

pytestmark = pytest.mark.gpu


class ThreadComm:
    """in-process stand-in for TorchComm: same interface, ranks are threads of one process on one GPU."""

    def __init__(self, rank, world, shared):
        self.rank, self.world, self.sh = rank, world, shared

    def all_reduce_sum_(self, t):
        sh = self.sh
        sh["sums"][self.rank] = t
        sh["bar"].wait()
        if self.rank == 0:
            import torch
            torch.cuda.synchronize()
            tot = sum(x.clone() for x in sh["sums"])
            torch.cuda.synchronize()
            sh["tot"] = tot
        sh["bar"].wait()
        t.copy_(sh["tot"])
        sh["bar"].wait()
        return t

    def exchange_wait(self, works):
        pass

    def exchange_start(self, send_lo, send_hi, recv_lo, recv_hi):
        import torch
        sh = self.sh
        sh["send"][self.rank] = (send_lo, send_hi)
        torch.cuda.synchronize()
        sh["bar"].wait()
        recv_lo.copy_(sh["send"][(self.rank - 1) % self.world][1])          # ring: strips are physical rows of a circular map
        recv_hi.copy_(sh["send"][(self.rank + 1) % self.world][0])
        torch.cuda.synchronize()
        sh["bar"].wait()
        return []


@pytest.mark.parametrize("world,cfg_name,C", [(2, "yaml", 130), (3, "default", 202), (2, "yaml_norays", 202), (4, "yaml", 130), (8, "default", 202),
                                               (4, "default_balanced", 202)])
def test_strip_contexts_reproduce_single_context(world, cfg_name, C, weights):
    import torch
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    from elevation_mapping_cupy_amd.sharded import ShardedElevationMap
    from _torch_strips import TorchStripEngine as HipStripEngine      # (the stage-by-stage orchestration needs exchange buffers: test infrastructure)
    from oracle import emap_oracle as eo
    from elevation_mapping_cupy_amd.sharded import ray_balanced_weights, strip_rows
    cfg = dict(eo.DEFAULTS); cfg.update(eo.YAML if cfg_name.startswith("yaml") else {})
    if cfg_name.endswith("norays"):
        cfg["enable_visibility_cleanup"] = False
    row_w = None
    if cfg_name.endswith("balanced"):      # strips of unequal height (equal ray work for a centred sensor, Parameter default ray length 2 m)
        row_w = ray_balanced_weights(C, cfg["resolution"], cfg["max_ray_length"], 6, world)
        heights = [strip_rows(C, world, r, row_w)[1] - strip_rows(C, world, r, row_w)[0] for r in range(world)]
        assert max(heights) > 2 * min(heights) and sum(heights) == C
    N = 40000
    R, t = fx.POSES["rotated"]
    clouds = [fx.cloud(C, N, f, dz=dz) for f, dz in enumerate((0.0, -0.02, -0.1))]
    full = ElevationMap(parameter_from(cfg, C, "reference_fp16", weights))
    for p in clouds:
        full.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0)
        for _ in range(6):
            full.update_time()
    want, want_n = full.elevation_map, full.normal_map
    dev = torch.device("cuda", 0)
    shared = {"bar": threading.Barrier(world), "sums": [None] * world, "send": [None] * world}
    out, errs = [None] * world, []

    def run(rank):
        try:
            eng = HipStripEngine(parameter_from(cfg, C, "reference_fp16", weights), rank, world, 0, dev, row_w)
            sm = ShardedElevationMap(eng, ThreadComm(rank, world, shared), cfg["enable_visibility_cleanup"], cfg["enable_overlap_clearance"])
            for p in clouds:
                eng.bind_points(p)
                sm.update(R, t, 1.0, 1.0)
                for _ in range(6):
                    eng.update_time()
            eng.sync()
            out[rank] = (eng.map.row_begin, eng.map.rows, eng.map.elevation_map, eng.map.normal_map, eng.map.get_additive_mean_error())
        except Exception as e:  # pragma: no cover
            errs.append(e)
            shared["bar"].abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [x.start() for x in th]; [x.join() for x in th]
    assert not errs, errs
    for r0, rows, m, nm, add in out:
        assert m.tobytes() == want[:, r0:r0 + rows].tobytes(), "strip at row %d differs" % r0
        assert nm.tobytes() == want_n[:, r0:r0 + rows].tobytes()
        assert add == full.get_additive_mean_error()


@pytest.mark.parametrize("world,scatter", [(2, "auto"), (3, "auto"), (2, "binned"), (3, "binned")])
def test_strips_fuse_rgb_and_semantic_channels_like_the_single_context(world, scatter, weights):
    """BASELINE config 5 on strips: the extra cloud channels are fused per cell into the strip's own layers (no exchange step);
    colour layer bit-exact, averaged layers bit-exact too (same LDS reduction per tile or the same atomics per cell)."""
    import torch
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    from elevation_mapping_cupy_amd.sharded import ShardedElevationMap
    from _torch_strips import TorchStripEngine as HipStripEngine      # (the stage-by-stage orchestration needs exchange buffers: test infrastructure)
    from oracle import emap_oracle as eo
    C, N = 202, 30000
    cfg = dict(eo.DEFAULTS); cfg.update(eo.YAML, enable_visibility_cleanup=False)
    CH = ["x", "y", "z", "s0", "s1", "c0", "rgb"]
    FUS = {"rgb": "color", "c0": "class_average", "default": "average"}
    R, t = fx.POSES["rotated"]
    clouds = [fx.semantic_cloud(C, N, f) for f in range(2)]

    def par():
        p = parameter_from(cfg, C, "reference_fp16", weights)
        p.pointcloud_channel_fusions = dict(FUS)
        return p
    full = ElevationMap(par())
    full.set_scatter_mode(scatter)
    for p in clouds:
        full.input_pointcloud(p, CH, R, t.copy() + full.center, 1.0, 1.0)
        full.update_time()
    want_e, want_s = full.elevation_map, full.semantic_map.semantic_map
    dev = torch.device("cuda", 0)
    shared = {"bar": threading.Barrier(world), "sums": [None] * world, "send": [None] * world}
    out, errs = [None] * world, []

    def run(rank):
        try:
            eng = HipStripEngine(par(), rank, world, 0, dev)
            eng.map.set_scatter_mode(scatter)          # binned: the strip point passes stage and carry the channels (emap_binned.hip)
            sm = ShardedElevationMap(eng, ThreadComm(rank, world, shared), False, cfg["enable_overlap_clearance"])
            for p in clouds:
                eng.bind_points(p)
                sm.update(R, t, 1.0, 1.0, CH)
                eng.update_time()
            eng.sync()
            out[rank] = (eng.map.row_begin, eng.map.rows, eng.map.elevation_map, eng.map.semantic_map.semantic_map)
        except Exception as e:  # pragma: no cover
            errs.append(e)
            shared["bar"].abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [x.start() for x in th]; [x.join() for x in th]
    assert not errs, errs
    for r0, rows, m, s in out:
        assert m.tobytes() == want_e[:, r0:r0 + rows].tobytes()
        assert s.shape == (4, rows, C) and s.tobytes() == want_s[:, r0:r0 + rows].tobytes()
    assert (want_s[3].view(np.uint32) != 0).sum() > 1000


def _rows_of(full_planes, begin, rows):
    """the strip's view: logical rows begin, begin + 1, ... (modulo cell_n) of the full map"""
    C = full_planes.shape[1]
    return np.take(full_planes, (begin + np.arange(rows)) % C, axis=1)


# row shifts of at most halo_rows (= 7) per move: this test drives the stages from Python (ShardedElevationMap over a thread "communicator"),
# which exchanges the normal planes' halo rows only and REFUSES a larger lag; the library's own frame (emap_update_sharded) fetches the
# rows from whoever owns them for any shift -- tests/test_hip_comm.py::test_strips_after_large_map_moves (DESIGN.md "Map shift")
MOVES = [(0.13, -0.3, 0.05), (-0.10, 0.49, -0.02), (0.17, 0.5, 0.0), (-0.05, -0.9, 0.11)]


@pytest.mark.parametrize("world,cfg_name", [(2, "yaml"), (3, "yaml"), (4, "yaml_norays")])
def test_strips_follow_the_robot_like_the_single_context(world, cfg_name, weights):
    """move_to between frames on row strips: every rank rotates its circular origin by the same amount (a strip keeps its physical
    rows, the logical rows it holds change, the halo ring hands the seam rows round); frames, decay and read-back must equal the
    single context bit for bit -- including the visibility pass, whose un-shifted normal planes are fetched from the neighbour."""
    import torch
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    from elevation_mapping_cupy_amd.sharded import ShardedElevationMap
    from _torch_strips import TorchStripEngine as HipStripEngine      # (the stage-by-stage orchestration needs exchange buffers: test infrastructure)
    from oracle import emap_oracle as eo
    C, N = 130, 30000
    cfg = dict(eo.DEFAULTS); cfg.update(eo.YAML)
    if cfg_name.endswith("norays"):
        cfg["enable_visibility_cleanup"] = False
    R, t0 = fx.POSES["rotated"]
    clouds = [fx.cloud(C, N, f, dz=-0.02 * f) for f in range(5)]

    def drive(frame, move, tick, center):
        for f, p in enumerate(clouds):
            frame(p, (t0 + center()).astype(np.float32))
            for _ in range(6):
                tick()
            if f < len(MOVES):
                move(np.array(MOVES[f], np.float64))
    full = ElevationMap(parameter_from(cfg, C, "reference_fp16", weights))
    drive(lambda p, tw: full.update_map_with_kernel(p, [], R, tw, 1.0, 1.0), lambda v: full.move_to(v, np.eye(3)), full.update_time, lambda: full.center)
    want, want_n = full.elevation_map, full.normal_map
    dev = torch.device("cuda", 0)
    shared = {"bar": threading.Barrier(world), "sums": [None] * world, "send": [None] * world}
    out, errs = [None] * world, []

    def run(rank):
        try:
            eng = HipStripEngine(parameter_from(cfg, C, "reference_fp16", weights), rank, world, 0, dev)
            sm = ShardedElevationMap(eng, ThreadComm(rank, world, shared), cfg["enable_visibility_cleanup"], cfg["enable_overlap_clearance"])

            def frame(p, tw):
                eng.bind_points(p)
                sm.update(R, (tw - eng.map.center).astype(np.float32), 1.0, 1.0)       # the strip protocol takes map-centre relative t
            drive(frame, lambda v: sm.move_to(v, np.eye(3)), eng.update_time, lambda: eng.map.center)
            eng.sync()
            out[rank] = (eng.map.logical_row_begin, eng.map.rows, eng.map.elevation_map, eng.map.normal_map)
        except Exception as e:  # pragma: no cover
            errs.append(e)
            shared["bar"].abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [x.start() for x in th]; [x.join() for x in th]
    assert not errs, errs
    assert sorted(b for b, *_ in out) != sorted((r * C) // world for r in range(world)), "the moves must have shifted rows"
    for b, rows, m, nm in out:
        assert m.tobytes() == _rows_of(want, b, rows).tobytes(), "strip whose view starts at logical row %d differs" % b
        assert nm.tobytes() == _rows_of(want_n, b, rows).tobytes()
