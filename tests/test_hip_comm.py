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
