"""GPU: the multi-GPU configurations of BASELINE.json as north_star states them -- configs[3] (4096 x 4096 over 4 ranks) and configs[4]
(8192 x 8192 multi-modal map, 16 M points per frame, 8 ranks) -- through the library's own sharded frame, emap_update_sharded:
count -> all-reduce -> gate folded into the tile kernel -> fuse -> [rays] -> halo exchange on the second stream next to the interior
stencil tiles -> boundary tiles, plus the per-strip RGB / semantic fusion.  One MI355X is available to the tests, so the ranks are
threads that drive their strip contexts on that GPU and the ten RCCL entry points are served by the stream-ordered in-process
stand-in (tests/fake_rccl/stream_rccl.hip: events across the ranks' streams, nothing synchronises with the host).

What these sizes exercise and the small strip tests do not: the STRIP variants of the sort front-end (cheap ownership test, per-wave
LDS queue compaction, staged records, block regions -- emap_binned.hip) with hundreds of blocks of tens of thousands of points,
bins of one tile on the strip where the single context sorts into bins of four, k_tile_semantic on a strip, the stencil kernels on
strips whose row pitch is 8192 cells, and 32-bit offsets into planes of more than a GiB.

Every strip must equal the rows of the single-context map BIT FOR BIT -- all seven planes, the normals, the semantic layers; the
single context is tied to the oracle at the same sizes by tests/test_hip_large_maps.py."""
import ctypes as ct
import threading

import numpy as np
import pytest

import _fixtures as fx
from _util import rccl_stand_in
from oracle import emap_oracle as eo

pytestmark = pytest.mark.gpu


class _DeviceClouds:
    """the replicated cloud, uploaded ONCE and bound by every context (a strip context of a real launch holds its own copy on its own GPU)"""

    def __init__(self, clouds):
        self.hip = ct.CDLL("libamdhip64.so")
        self.ptrs, self.n, self.stride = [], clouds[0].shape[0], clouds[0].shape[1]
        for p in clouds:
            p = np.ascontiguousarray(p, np.float32)
            d = ct.c_void_p()
            assert self.hip.hipMalloc(ct.byref(d), ct.c_size_t(p.nbytes)) == 0
            assert self.hip.hipMemcpy(d, ct.c_void_p(p.ctypes.data), ct.c_size_t(p.nbytes), 1) == 0
            self.ptrs.append(d)

    def free(self):
        for d in self.ptrs:
            self.hip.hipFree(d)


def _sharded_vs_single(world, cfg, C, clouds, channels, fusions, ticks, weights, stand_in="stream", ray_mode="auto"):
    import torch
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    from elevation_mapping_cupy_amd.sharded import HipStripEngine, NativeComm, ShardedElevationMap
    lib_path = rccl_stand_in(stand_in)
    dc = _DeviceClouds(clouds)
    extra = list(channels[3:]) if channels else []
    R, t = fx.POSES["rotated"]

    def par():
        p = parameter_from(cfg, C, "fp32", weights)
        if fusions:
            p.pointcloud_channel_fusions = dict(fusions)
        return p
    full = ElevationMap(par())
    for d in dc.ptrs:
        full.bind_points_device(d.value, dc.n, dc.stride)
        full.update_map_with_kernel(None, extra, R, t.copy(), 1.0, 1.0, want_stats=False)
        for _ in range(ticks):
            full.update_time()
    full.sync()
    want_add = full.get_additive_mean_error()
    uid = (ct.c_uint8 * 128)()
    assert full._lib.emap_comm_unique_id(lib_path.encode(), uid) == 0
    dev = torch.device("cuda", 0)
    engs, errs = [None] * world, []

    def run(rank):
        try:
            eng = HipStripEngine(par(), rank, world, 0, dev)
            eng.map.set_ray_mode(ray_mode)
            comm = NativeComm(eng, rank=rank, world=world, bootstrap=False, uid=bytes(uid), rccl_path=lib_path)
            assert comm.rccl_ranks() == world
            sm = ShardedElevationMap(eng, comm, cfg["enable_visibility_cleanup"], cfg["enable_overlap_clearance"])
            for d in dc.ptrs:
                eng.bind_points_device(d.value, dc.n, dc.stride)
                sm.update(R, t, 1.0, 1.0, channels)
                for _ in range(ticks):
                    eng.update_time()
            eng.sync()
            engs[rank] = eng
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    [x.start() for x in th]
    [x.join(timeout=600) for x in th]
    assert not any(x.is_alive() for x in th), "a rank is stuck in the exchange"
    assert not errs, errs
    # plane by plane (268 MB each at 8192^2): the full map's plane against every strip's rows
    names = list(full.layer_names_core) + ["normal_x", "normal_y", "normal_z"]
    n_valid = 0
    for k, name in enumerate(names):
        want = full.get_layer_raw(k)
        if k == 2:
            n_valid = int((want > 0.5).sum())
        for eng in engs:
            r0, rows = eng.map.row_begin, eng.map.rows
            got = eng.map.get_layer_raw(k)
            assert got.tobytes() == want[r0:r0 + rows].tobytes(), "plane %s of the strip at row %d differs from the single context (%d cells)" % (
                name, r0, int((got.view(np.uint32) != want[r0:r0 + rows].view(np.uint32)).sum()))
    sem_names = list(full.semantic_map.layer_names)
    for name in sem_names:
        want = full.semantic_map.get_layer(name)
        for eng in engs:
            r0, rows = eng.map.row_begin, eng.map.rows
            assert eng.map.semantic_map.layer_names == sem_names
            got = eng.map.semantic_map.get_layer(name)
            assert got.tobytes() == want[r0:r0 + rows].tobytes(), "semantic layer %s of the strip at row %d differs" % (name, r0)
        if name == "rgb":
            assert int((want.view(np.uint32) != 0).sum()) > n_valid // 2
    for eng in engs:
        assert eng.map.get_additive_mean_error() == want_add
        eng.lib.emap_comm_destroy(eng.ctx)
        eng.map.close()
    full.close()
    dc.free()
    return n_valid


@pytest.mark.parametrize("rays,world", [(None, 4), ("by_row", 4), ("by_ray", 4), ("by_ray", 8)])
def test_config4_4096_sharded_frame(rays, world, weights):
    """BASELINE configs[3]: 4096^2 over 4 ranks (and 8).  Without the visibility pass the strips run the STRIP sort variants at size (4 M
    points, 253 blocks).  With it: BY ROW every valid point marches its ray through the strip (ray-only bin, the non-strip sort
    kernels); BY RAY -- what such a map selects by itself -- every rank sorts and marches only the points of its rows over the
    all-reduced 520 x 576-cell window around the sensor, and the effects come back through two integer all-reduces."""
    C, N = 4096, 4_000_000
    cfg = dict(eo.DEFAULTS); cfg.update(eo.YAML, enable_visibility_cleanup=rays is not None)
    clouds = [fx.cloud(C, N, 90 + f, dz=dz) for f, dz in enumerate((0.0, -0.12))]
    n_valid = _sharded_vs_single(world, cfg, C, clouds, None, None, 8, weights, ray_mode=rays or "auto")
    assert n_valid > 1_500_000


def test_config5_8192_multimodal_over_8_ranks_sharded_frame(weights):
    """BASELINE configs[4] as north_star states it: 8192^2 multi-modal map (height + RGB + 3 semantic layers), 16 M points per frame,
    8 ranks with the halo exchange overlapped with the interior stencils; two frames (the second one fuses into a warm map)."""
    C, N = 8192, 16_000_000
    cfg = dict(eo.DEFAULTS); cfg.update(eo.YAML, enable_visibility_cleanup=False, enable_overlap_clearance=False)
    clouds = []
    for f, dz in enumerate((0.0, -0.04)):
        p = fx.cloud(C, N, 60 + f, dz=dz, extra=4)
        p[:, 3] = np.random.default_rng(70 + f).integers(0, 1 << 24, N, dtype=np.uint32).view(np.float32)
        p[::5, :2] = p[1::5, :2][: p[::5].shape[0]]                      # pile points up: cells with several points
        clouds.append(p)
    n_valid = _sharded_vs_single(8, cfg, C, clouds, ["x", "y", "z", "rgb", "s0", "s1", "s2"], {"rgb": "color", "default": "average"}, 3, weights)
    assert n_valid > 5_000_000
