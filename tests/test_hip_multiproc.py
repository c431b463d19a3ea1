"""GPU, multi-process: two OS processes (one rank each) drive two strips on the SAME MI355X, the frame orchestrated stage by stage
(ShardedElevationMap._update) -- gloo carries the collectives because RCCL refuses two ranks on one device and the in-process RCCL
stand-ins cannot span processes; communicator and buffer-owning engine are test infrastructure (tests/_torch_strips.py).  The
product's own multi-rank frame (emap_update_sharded over RCCL) is covered by test_hip_comm.py / test_hip_large_strips.py."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, outdir, C, N, rays):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch
    import torch.distributed as dist
    import _fixtures as fx
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.sharded import ShardedElevationMap
    from _torch_strips import TorchComm, TorchStripEngine as HipStripEngine
    from oracle import emap_oracle as eo
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = dict(eo.DEFAULTS); cfg.update(eo.YAML); cfg["enable_visibility_cleanup"] = rays
    w = np.load(os.path.join(ROOT, "tests", "golden", "weights.npz")); weights = {k: w[k] for k in w.files}
    eng = HipStripEngine(parameter_from(cfg, C, "reference_fp16", weights), rank, world, 0, dev)
    sm = ShardedElevationMap(eng, TorchComm(dev), cfg["enable_visibility_cleanup"], cfg["enable_overlap_clearance"])
    R, t = fx.POSES["rotated"]
    for f, dz in enumerate((0.0, -0.02, -0.1)):
        eng.bind_points(fx.cloud(C, N, f, dz=dz))
        sm.update(R, t, 1.0, 1.0)
        for _ in range(6):
            eng.update_time()
    eng.sync()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), r0=eng.map.row_begin, rows=eng.map.rows, emap=eng.map.elevation_map,
             normal=eng.map.normal_map, add_err=np.float32(eng.map.get_additive_mean_error()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("rays", [True, False])
def test_two_processes_two_strips_one_gpu(rays, weights):
    import torch.multiprocessing as mp
    import _fixtures as fx
    from elevation_mapping_cupy_amd.configs import parameter_from
    from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
    from oracle import emap_oracle as eo
    C, N, world = 202, 40000, 2
    outdir = tempfile.mkdtemp()
    mp.spawn(_worker, args=(world, _free_port(), outdir, C, N, rays), nprocs=world, join=True)
    cfg = dict(eo.DEFAULTS); cfg.update(eo.YAML); cfg["enable_visibility_cleanup"] = rays
    full = ElevationMap(parameter_from(cfg, C, "reference_fp16", weights))
    R, t = fx.POSES["rotated"]
    for f, dz in enumerate((0.0, -0.02, -0.1)):
        full.update_map_with_kernel(fx.cloud(C, N, f, dz=dz), [], R, t.copy(), 1.0, 1.0)
        for _ in range(6):
            full.update_time()
    want, want_n = full.elevation_map, full.normal_map
    for r in range(world):
        g = np.load(os.path.join(outdir, "rank%d.npz" % r))
        r0, rows = int(g["r0"]), int(g["rows"])
        assert g["emap"].tobytes() == want[:, r0:r0 + rows].tobytes(), "rank %d" % r
        assert g["normal"].tobytes() == want_n[:, r0:r0 + rows].tobytes()
        assert float(g["add_err"]) == full.get_additive_mean_error()
