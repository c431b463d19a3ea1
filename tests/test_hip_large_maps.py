"""GPU: parity at the sizes of BASELINE.json configs[3] and configs[4] -- where the large-map code paths live:

  * config 4: 4096 x 4096 map, 4 M points, `fp32` index mode (the reference's half-precision helpers cannot address more than 2049
    cells per axis), visibility rays + overlap clearance on.  The inert bitmap no longer fits LDS (2 MB), the ray pass takes its
    global-memory path; 16384 sort bins.  Single context vs the OpenMP oracle, and 4 row-strip contexts vs the single context.
  * config 5: 8192 x 8192 multi-modal map (height + RGB + 3 semantic layers), 16 M points: bins of 4 stacked tiles (sub = 4,
    emap_binned.hip), k_tile_semantic at size, 67 M cells (32-bit offsets into 16-byte planes of > 1 GiB).

Indices / flags bit-exact, every core plane and the normals BIT FOR BIT (north_star asks for 1e-5), colour layer bit-exact, averaged
semantic layers within 1e-6.  These tests need ~6 GB of host memory
and a few tens of seconds of oracle time each; they stay inside `-m gpu`."""
import threading

import numpy as np
import pytest

import _fixtures as fx
from _util import assert_planes_close, assert_planes_equal
from oracle import emap_oracle as eo

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _oracle_threads():
    eo.set_threads(16)
    yield
    eo.set_threads(1)


def _cfg4():
    cfg = dict(eo.DEFAULTS); cfg.update(eo.YAML)       # rays + overlap clearance on, shipped core_param.yaml values
    return cfg


def test_config4_4096_fp32_rays_vs_oracle(weights):
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    C, N = 4096, 4_000_000
    cfg = _cfg4()
    hip = ElevationMap(parameter_from(cfg, C, "auto", weights))
    assert hip.index_mode == "fp32"
    orc = eo.OracleMap(eo.make_params(cfg, cell_n=C, mode="fp32", weights=weights))
    R, t = fx.POSES["rotated"]
    visits = 0
    for f, dz in enumerate((0.0, -0.15, -0.05)):
        p = fx.cloud(C, N, 40 + f, dz=dz)
        hip.bind_points(p)
        if f == 0:
            i1 = hip.point_index(R, t); i0 = orc.point_index(p, R, t)
            assert all(np.array_equal(a, b) for a, b in zip(i1, i0)), "cell indices / flags must be bit-exact"
            assert int((i1[1] & i1[2]).sum()) > N // 4
        hip.update_map_with_kernel(None, [], R, t.copy(), 1.0, 1.0)
        orc.update_map_with_kernel(p, R, t, 1.0, 1.0)
        visits += orc.last["ray_visits"]
        for _ in range(8):
            hip.update_time(); orc.update_time()
        hip.update_variance(); orc.update_variance()
    assert visits > 3e8                                           # the ray pass really ran at size
    m = hip.elevation_map
    assert np.array_equal(m[2] > 0.5, orc.elevation_map[2] > 0.5), "validity flags must be exact"
    assert np.array_equal(m[6], orc.elevation_map[6])
    assert np.array_equal(m[3], orc.elevation_map[3]), "traversability must be bit-exact"
    assert_planes_close(m, orc.elevation_map, what="config 4, 3 frames")
    assert_planes_equal(m, orc.elevation_map, what="config 4, 3 frames")
    assert_planes_equal(hip.normal_map, orc.normal_map, names=["nx", "ny", "nz"])
    assert hip.get_additive_mean_error() == float(orc.additive_mean_error)


def test_config4_four_strips_reproduce_the_single_context(weights):
    """4 row-strip contexts of the 4096^2 map on one device (threads; the exchange steps go through the strips' C-ABI entry points
    exactly as under RCCL) == the single context, bit for bit -- with rays, i.e. the band-limited march of k_rays<STRIP> at size."""
    import torch
    from test_hip_strips import ThreadComm
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    from elevation_mapping_cupy_amd.sharded import ShardedElevationMap
    from _torch_strips import TorchStripEngine as HipStripEngine      # (the stage-by-stage orchestration needs exchange buffers: test infrastructure)
    C, N, world = 4096, 2_000_000, 4
    cfg = _cfg4()
    R, t = fx.POSES["rotated"]
    clouds = [fx.cloud(C, N, 50 + f, dz=dz) for f, dz in enumerate((0.0, -0.12))]
    full = ElevationMap(parameter_from(cfg, C, "fp32", weights))
    for p in clouds:
        full.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0)
        for _ in range(8):
            full.update_time()
    want, want_n = full.elevation_map, full.normal_map
    add = full.get_additive_mean_error()
    full.close()
    dev = torch.device("cuda", 0)
    shared = {"bar": threading.Barrier(world), "sums": [None] * world, "send": [None] * world}
    out, errs = [None] * world, []

    def run(rank):
        try:
            eng = HipStripEngine(parameter_from(cfg, C, "fp32", weights), rank, world, 0, dev)
            sm = ShardedElevationMap(eng, ThreadComm(rank, world, shared), True, True)
            for p in clouds:
                eng.bind_points(p)
                sm.update(R, t, 1.0, 1.0)
                for _ in range(8):
                    eng.update_time()
            eng.sync()
            out[rank] = (eng.map.row_begin, eng.map.rows, eng.map.elevation_map, eng.map.normal_map, eng.map.get_additive_mean_error())
        except Exception as e:  # pragma: no cover
            errs.append(e)
            shared["bar"].abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [x.start() for x in th]; [x.join() for x in th]
    assert not errs, errs
    for r0, rows, m, nm, a in out:
        assert m.tobytes() == want[:, r0:r0 + rows].tobytes(), "strip at row %d differs" % r0
        assert nm.tobytes() == want_n[:, r0:r0 + rows].tobytes()
        assert a == add


def test_config5_8192_multimodal_vs_oracle(weights):
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    C, N = 8192, 16_000_000
    cfg = dict(eo.DEFAULTS); cfg.update(eo.YAML, enable_visibility_cleanup=False, enable_overlap_clearance=False)
    par = parameter_from(cfg, C, "auto", weights)
    par.pointcloud_channel_fusions = {"rgb": "color", "default": "average"}
    hip = ElevationMap(par)
    assert hip.index_mode == "fp32"
    orc = eo.OracleMap(eo.make_params(cfg, cell_n=C, mode="fp32", weights=weights))
    CH = ["x", "y", "z", "rgb", "s0", "s1", "s2"]
    R, t = fx.POSES["rotated"]
    for f, dz in enumerate((0.0, -0.04)):
        p = fx.cloud(C, N, 60 + f, dz=dz, extra=4)
        p[:, 3] = np.random.default_rng(70 + f).integers(0, 1 << 24, N, dtype=np.uint32).view(np.float32)
        p[::5, :2] = p[1::5, :2][: p[::5].shape[0]]                      # pile points up: cells with several points
        hip.input_pointcloud(p, CH, R, t.copy(), 1.0, 1.0)
        if f == 0:
            i1 = hip.point_index(R, t); i0 = orc.point_index(p, R, t)
            assert all(np.array_equal(a, b) for a, b in zip(i1, i0)), "cell indices / flags must be bit-exact"
        orc.update_map_with_kernel(p, R, t, 1.0, 1.0)
        orc.semantic_update(p, R, t, average=[(4, 1), (5, 2), (6, 3)], color=[(3, 0)])
        for _ in range(3):
            hip.update_time(); orc.update_time()
    assert hip.semantic_map.layer_names == ["rgb", "s0", "s1", "s2"]
    for k in range(7):                                                   # plane by plane: 268 MB each
        a = hip.get_layer_raw(k)
        assert_planes_close(a[None], orc.elevation_map[k][None], names=[hip.layer_names_core[k]], what="config 5")
        assert_planes_equal(a[None], orc.elevation_map[k][None], names=[hip.layer_names_core[k]], what="config 5")
    for j, k in enumerate((7, 8, 9)):
        assert hip.get_layer_raw(k).tobytes() == orc.normal_map[j].tobytes(), "normal plane %d" % j
    assert np.array_equal(hip.semantic_map.get_layer("rgb").view(np.uint32), orc.semantic_map[0].view(np.uint32)), "colour layer must be bit-exact"
    for j, name in enumerate(("s0", "s1", "s2")):
        assert np.allclose(hip.semantic_map.get_layer(name), orc.semantic_map[1 + j], atol=1e-6, rtol=1e-5), name
    assert int((orc.elevation_map[2] > 0.5).sum()) > 5_000_000


def test_sort_offsets_under_stress(weights):
    """The counting sort's offsets come from a last-workgroup hand-off that is ordered by device-coherent stores + s_waitcnt instead of a
    release / acquire pair (emap_device.h: last_block_ticket).  If that ordering ever broke, tile_start would be stale and records
    would land in the wrong tiles -- silently.  40 frames with 16384 sort bins (4096^2 map), fresh clouds every frame: the map of
    the tile-binned scatter must equal the map of the global-atomic scatter bit for bit after every tenth frame."""
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    C, N = 4096, 1_500_000
    cfg = dict(eo.DEFAULTS); cfg.update(eo.YAML, enable_visibility_cleanup=False)
    maps = []
    for mode in ("binned", "atomic"):
        m = ElevationMap(parameter_from(cfg, C, "fp32", weights))
        m.set_scatter_mode(mode)
        maps.append(m)
    R, t = fx.POSES["identity"]
    rng = np.random.default_rng(99)
    base = fx.cloud(C, N, 80)
    for f in range(40):
        p = base.copy()
        p[:, :2] += rng.uniform(-0.3, 0.3, 2).astype(np.float32)       # a new cloud every frame, cheaply
        p[:, 2] += np.float32(0.01 * (f % 7))
        for m in maps:
            m.update_map_with_kernel(p, [], R, t.copy(), 1.0, 1.0, want_stats=False)
        if f % 10 == 9:
            for k in (0, 1, 2, 5):
                a, b = maps[0].get_layer_raw(k), maps[1].get_layer_raw(k)
                assert a.tobytes() == b.tobytes(), "frame %d, plane %d: binned and atomic scatter disagree" % (f, k)
