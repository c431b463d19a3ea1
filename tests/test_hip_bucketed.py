"""GPU: clouds BUCKETED per rank (round 5).  Every rank of a sharded map is handed the same sensor cloud; where the frame allows it
a rank converts, uploads and streams only the points that can land in its rows (include/emap_hip.h: emap_upload_points_strip,
ShardedElevationMap.input_pointcloud) -- the host-side test is a conservative superset of what the kernels then decide exactly, so the
strips must still equal the rows of the single-context map BIT FOR BIT.  Reference side: input_pointcloud uploads the whole cloud
(EM/elevation_mapping.py:434-466); SURVEY 8(e): "host bucket by strip"."""
import ctypes as ct
import threading

import numpy as np
import pytest

import _fixtures as fx
from _util import rccl_stand_in
from oracle import emap_oracle as eo

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("C,world,mode,pose", [(300, 4, "reference_fp16", "rotated"), (1024, 8, "reference_fp16", "rotated"), (421, 3, "fp32", "identity"), (2048, 8, "reference_fp16", "rotated")])
def test_the_host_side_test_keeps_every_point_the_kernels_act_on(C, world, mode, pose, weights):
    """superset property against the EXACT cell indices of the single context (emap_point_index: the kernels' own arithmetic), on a
    map whose circular origin has moved; and it really buckets: a rank keeps about 1 / world of a uniform cloud"""
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    from elevation_mapping_cupy_amd.sharded import halo_rows_needed, strip_rows
    cfg = dict(eo.DEFAULTS); cfg.update(eo.YAML)
    R, t = fx.POSES[pose]
    full = ElevationMap(parameter_from(cfg, C, mode, weights))
    strips = []
    for r in range(world):
        r0, r1 = strip_rows(C, world, r, None)
        strips.append(ElevationMap(parameter_from(cfg, C, mode, weights), strip=(r0, r1 - r0, halo_rows_needed(cfg["dilation_size"], world))))
    rng = np.random.default_rng(C)
    for step, mv in enumerate([None, (0.53, -0.3, 0.0), (-2.1, 0.4, 0.0)]):
        if mv is not None:
            for m in [full] + strips:
                m.move_to(np.array(mv, np.float64), np.eye(3))
        p = fx.cloud(C, 120_000, 70 + step)
        p[::97] *= 1.6                                     # points beyond the map: clamped to its first / last row (ray-only points there)
        p[::1013, int(rng.integers(0, 3))] = np.nan        # NaN rows: skipped by every kernel, never uploaded
        t_rel = (t + full.center).astype(np.float32) - full.center
        full.bind_points(p)
        idx, valid, inside = full.point_index(R, t_rel)
        row = np.asarray(idx) // C                         # exact LOGICAL row of every point (cell index = C * ix + iy)
        finite = ~np.isnan(p).any(axis=1)
        kept_total = 0
        for s in strips:
            keep = s.strip_point_mask(p, R, t_rel)
            b = s.logical_row_begin
            owned = ((row - b) % C) < s.rows
            must = owned & finite & (np.asarray(valid) > 0)
            assert not np.any(must & ~keep), "%d points the strip's kernels act on were not kept" % int((must & ~keep).sum())
            assert not np.any(keep & ~finite), "NaN rows must not be uploaded"
            kept_total += int(keep.sum())
            assert keep.sum() <= p.shape[0] * (1.0 / world + 0.08) + 0.02 * p.shape[0], "a strip kept %d of %d points" % (int(keep.sum()), p.shape[0])
        assert kept_total >= int((finite & (np.asarray(valid) > 0)).sum())
    for m in [full] + strips:
        m.close()


def _bucketed_vs_single(world, cfg, C, frames, weights, mode="reference_fp16", stand_in="stream", ray_mode="auto", channels=None, fusions=None):
    """frames = [(cloud, R, t, n_update_time, move_to or None)]: the single context through input_pointcloud, every rank of the sharded
    map through ShardedElevationMap.input_pointcloud with the SAME (whole) cloud; strips == rows of the single map, bit for bit.
    Returns the largest share of a cloud any rank bound."""
    import torch
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    from elevation_mapping_cupy_amd.sharded import HipStripEngine, NativeComm, ShardedElevationMap
    lib_path = rccl_stand_in(stand_in)
    names = ["x", "y", "z"] + list(channels or [])

    def par():
        p = parameter_from(cfg, C, mode, weights)
        if fusions:
            p.pointcloud_channel_fusions = dict(fusions)
        return p
    full = ElevationMap(par())
    t_rel = []
    for p, R, t, nt, mv in frames:
        tw = (t + full.center).astype(np.float32)
        t_rel.append(tw - full.center)
        full.input_pointcloud(p, names, R, tw, 1.0, 1.0)
        for _ in range(nt):
            full.update_time()
        if mv is not None:
            full.move_to(np.array(mv, np.float64), np.eye(3))
    want, want_n, want_add = full.elevation_map, full.normal_map, full.get_additive_mean_error()
    want_sem = {n: full.semantic_map.get_layer(n) for n in full.semantic_map.layer_names} if channels else {}
    uid = (ct.c_uint8 * 128)()
    assert full._lib.emap_comm_unique_id(lib_path.encode(), uid) == 0
    dev = torch.device("cuda", 0)
    out, errs, share = [None] * world, [], [0.0] * world

    def run(rank):
        try:
            eng = HipStripEngine(par(), rank, world, 0, dev)
            eng.map.set_scatter_mode("binned")
            eng.map.set_ray_mode(ray_mode)
            comm = NativeComm(eng, rank=rank, world=world, bootstrap=False, uid=bytes(uid), rccl_path=lib_path)
            sm = ShardedElevationMap(eng, comm, cfg["enable_visibility_cleanup"], cfg["enable_overlap_clearance"])
            for (p, R, _t, nt, mv), t in zip(frames, t_rel):
                assert sm.buckets_clouds(p.shape[0], names)
                sm.input_pointcloud(p, names, R, t, 1.0, 1.0)
                share[rank] = max(share[rank], eng.map._n_bound / p.shape[0])
                for _ in range(nt):
                    eng.update_time()
                if mv is not None:
                    sm.move_to(np.array(mv, np.float64), np.eye(3))
            eng.sync()
            sem = {n: eng.map.semantic_map.get_layer(n) for n in eng.map.semantic_map.layer_names} if channels else {}
            out[rank] = (eng.map.logical_row_begin, eng.map.rows, eng.map.elevation_map, eng.map.normal_map, eng.map.get_additive_mean_error(), sem, eng.map.row_begin)
            eng.lib.emap_comm_destroy(eng.ctx)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    [x.start() for x in th]
    [x.join(timeout=300) for x in th]
    assert not any(x.is_alive() for x in th), "a rank is stuck in the exchange"
    assert not errs, errs
    for b, rows, m, nm, add, sem, pb in out:
        idx = (b + np.arange(rows)) % C
        assert m.tobytes() == np.take(want, idx, axis=1).tobytes(), "strip whose view starts at logical row %d differs" % b
        assert nm.tobytes() == np.take(want_n, idx, axis=1).tobytes(), "normals of the strip at logical row %d differ" % b
        assert add == want_add
        for n, plane in sem.items():          # semantic layers are stored in physical rows on both sides (no move in the multi-modal case)
            assert plane.tobytes() == want_sem[n][pb:pb + rows].tobytes(), "semantic layer %s differs" % n
    return max(share)


@pytest.mark.parametrize("world,cfg_name,C,N,mode,ray_mode,stand_in,moves", [
    (2, "yaml_norays", 300, 150_000, "reference_fp16", "auto", "blocking", False),
    (4, "yaml_norays", 1024, 400_000, "reference_fp16", "auto", "stream", True),
    (8, "yaml_norays", 1024, 400_000, "reference_fp16", "auto", "stream", False),
    (4, "yaml", 300, 150_000, "reference_fp16", "by_ray", "stream", True),
    (8, "yaml", 421, 200_000, "fp32", "by_ray", "blocking", True),
    (3, "default", 300, 150_000, "reference_fp16", "by_ray", "stream", True)])
def test_bucketed_uploads_reproduce_the_single_context(world, cfg_name, C, N, mode, ray_mode, stand_in, moves, weights):
    cfg = dict(eo.DEFAULTS); cfg.update(eo.YAML if cfg_name.startswith("yaml") else {})
    cfg["enable_visibility_cleanup"] = not cfg_name.endswith("norays")
    R, t = fx.POSES["rotated"]
    MV = [(0.13, -0.3, 0.05), (1.2, 0.17, -0.02), None] if moves else [None] * 3      # (the second move: 30 rows)
    frames = [(fx.cloud(C, N, 40 + f, dz=dz), R, t + np.array([0.3 * f, -0.2 * f, 0], np.float32), 6, mv) for (f, dz), mv in zip(enumerate((0.0, -0.02, -0.1)), MV)]
    share = _bucketed_vs_single(world, cfg, C, frames, weights, mode=mode, stand_in=stand_in, ray_mode=ray_mode)
    assert share < 1.0 / world + 0.1, "the largest share a rank uploaded: %.3f" % share


def test_bucketed_multimodal_cloud_on_4_strips(weights):
    """RGB + averaged semantic channels ride with the bucketed points (de-interleaved on the way, like every uploaded cloud)"""
    C, N = 1024, 400_000
    cfg = dict(eo.DEFAULTS); cfg.update(eo.YAML, enable_visibility_cleanup=False)
    R, t = fx.POSES["rotated"]
    clouds = []
    for s in range(2):
        p = fx.cloud(C, N, 60 + s, dz=-0.03 * s, extra=4)
        p[:, 3] = np.random.default_rng(5 + s).integers(0, 1 << 24, N, dtype=np.uint32).view(np.float32)
        clouds.append(p)
    frames = [(p, R, t, 3, None) for p in clouds]
    share = _bucketed_vs_single(4, cfg, C, frames, weights, channels=["rgb", "sem0", "sem1", "sem2"], fusions={"rgb": "color", "default": "average"})
    assert share < 0.35


def test_a_bucketed_cloud_refuses_what_it_cannot_serve(weights):
    """a frame with another pose, and a visibility pass that marches by row, must fail loudly instead of fusing a partial cloud"""
    import torch
    from elevation_mapping_cupy_amd._lib import EmapError
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.sharded import HipStripEngine, NativeComm, ShardedElevationMap
    cfg = dict(eo.DEFAULTS); cfg.update(eo.YAML)
    lib_path = rccl_stand_in("blocking")
    R, t = fx.POSES["rotated"]
    errs = []
    uid = (ct.c_uint8 * 128)()
    from elevation_mapping_cupy_amd import _lib
    assert _lib.load().emap_comm_unique_id(lib_path.encode(), uid) == 0

    def run(rank):
        try:
            eng = HipStripEngine(parameter_from(cfg, 300, "reference_fp16", weights), rank, 2, 0, torch.device("cuda", 0))
            eng.map.set_scatter_mode("binned")
            comm = NativeComm(eng, rank=rank, world=2, bootstrap=False, uid=bytes(uid), rccl_path=lib_path)
            sm = ShardedElevationMap(eng, comm, True, True)
            p = fx.cloud(300, 150_000, 1)
            assert not sm.buckets_clouds(p.shape[0], ["x", "y", "z"])          # 300^2: rays march by row -> input_pointcloud uploads the whole cloud
            eng.map.bind_points(p, strip_pose=(R, t))                           # ... forcing a bucketed one under it is refused by the library
            with pytest.raises(EmapError, match="BY ROW"):
                sm.update(R, t, 1.0, 1.0)
            eng.map.set_ray_mode("by_ray")
            with pytest.raises(EmapError, match="another pose"):
                sm.update(R, t + np.float32(0.25), 1.0, 1.0)
            sm.update(R, t, 1.0, 1.0)                                            # the pose it was bucketed for: fine (collective: both ranks get here)
            eng.sync()
            # a ROW shift moves the strip's logical rows under the kept points: the same pose is refused too (every rank shifts alike),
            # by the frame and by the staged entry point, until the cloud is bucketed again
            eng.map.shift_map_xy(np.array([5, 0]))
            with pytest.raises(EmapError, match="rows shifted"):
                sm.update(R, t, 1.0, 1.0)
            with pytest.raises(EmapError, match="rows shifted"):
                eng.map.stage("count", R, t)
            eng.map.bind_points(p, strip_pose=(R, t))
            sm.update(R, t, 1.0, 1.0)
            eng.sync()
            eng.lib.emap_comm_destroy(eng.ctx)
        except Exception as e:  # pragma: no cover
            errs.append(e)
    th = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(2)]
    [x.start() for x in th]
    [x.join(timeout=120) for x in th]
    assert not any(x.is_alive() for x in th) and not errs, errs
