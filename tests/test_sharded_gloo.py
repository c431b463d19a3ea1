"""CPU, multi-process: the row-strip orchestration of elevation_mapping_cupy_amd/sharded.py (strip layout, the
drift all-reduce, halo width and exchange order) driven over torch.distributed/gloo with world_size 2 and 3.
The local compute of each rank is the CPU oracle restricted to its strip (tests may use the oracle; on the GPU box
the same orchestration runs with HipStripEngine).  The union of the owned rows must equal a single-process run."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class OracleStripEngine:
    def __init__(self, cfg, C, rank, world, weights):
        import torch
        from elevation_mapping_cupy_amd.sharded import halo_rows_needed, strip_rows
        from oracle import emap_oracle as eo
        self.torch, self.eo = torch, eo
        self.om = eo.OracleMap(eo.make_params(cfg, cell_n=C, weights=weights))
        self.r0, self.r1 = strip_rows(C, world, rank)
        self.H = halo_rows_needed(self.om.P.dilation_size, world)
        eo.lib().eo_set_strip(self.r0, self.r1)
        self.C = C
        # poison the rows this rank does not own: only the halo exchange may make them usable
        rng = np.random.default_rng(100 + rank)
        mask = np.ones(C, bool); mask[self.r0:self.r1] = False
        self.om.elevation_map[:, mask, :] = rng.uniform(-5, 5, (7, int(mask.sum()), C)).astype(np.float32)
        self.recv = [torch.zeros((7, max(self.H, 1), C), dtype=torch.float32) for _ in range(2)]
        # what ShardedElevationMap.gather reads from an engine: the strip's first logical row and its rows of a plane
        import types
        self.map = types.SimpleNamespace(logical_row_begin=self.r0, get_layer_raw=lambda pid: self.om.elevation_map[pid, self.r0:self.r1].copy())

    def bind_points(self, p):
        self.p = p

    def count(self, R, t):
        self.om.count(self.p, R, t)

    def local_sums(self):
        return self.torch.tensor([self.om.last["err_sum"], float(self.om.last["err_cnt"])], dtype=self.torch.float64)

    def gate(self, pn, on, totals):
        self.om.last["err_sum"], self.om.last["err_cnt"] = float(totals[0]), int(round(float(totals[1])))
        self.om.gate(pn, on)

    def fuse(self, R, t): self.om.fuse(self.p, R, t)
    def fuse_average(self, R, t): self.om.fuse(self.p, R, t); self.om.commit(); self.om.average()
    def commit(self): self.om.commit()
    def rays(self, R, t): self.om.rays(self.p, R, t)
    def average(self): self.om.average()
    def overlap(self, tz): self.om.overlap_clear(tz)

    def halo_pack(self):
        m, H = self.om.elevation_map, self.H
        lo = self.torch.from_numpy(np.ascontiguousarray(m[:, self.r0:self.r0 + H]))
        hi = self.torch.from_numpy(np.ascontiguousarray(m[:, self.r1 - H:self.r1]))
        return lo, hi, self.recv[0], self.recv[1]

    def halo_unpack(self, have_lo, have_hi):
        m, H = self.om.elevation_map, self.H
        # the exchange is a ring (strips are physical rows of a circular map); this engine's map is not shifted, so the rows that
        # arrive across the wrap (rank 0's lower, the last rank's upper halo) lie beyond the logical map border and are dropped
        if have_lo and self.r0 > 0:
            m[:, self.r0 - H:self.r0] = self.recv[0].numpy()
        if have_hi and self.r1 < self.C:
            m[:, self.r1:self.r1 + H] = self.recv[1].numpy()

    def post(self, part=0):
        if part != 1:          # the oracle has no tile split: everything happens in the "boundary" call
            self.om.dilate(); self.om.traversability(); self.om.normals()
    def update_time(self): self.om.update_time()


def _worker(rank, world, port, outdir, cfg_name, C, N):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    import _fixtures as fx
    from elevation_mapping_cupy_amd.sharded import ShardedElevationMap
    from _torch_strips import TorchComm
    from oracle import emap_oracle as eo
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = dict(eo.YAML if cfg_name.startswith("yaml") else eo.DEFAULTS)
    if cfg_name.endswith("norays"):
        cfg["enable_visibility_cleanup"] = False
    w = np.load(os.path.join(ROOT, "tests", "golden", "weights.npz")); weights = {k: w[k] for k in w.files}
    eng = OracleStripEngine(cfg, C, rank, world, weights)
    sm = ShardedElevationMap(eng, TorchComm(None), cfg["enable_visibility_cleanup"], cfg["enable_overlap_clearance"])
    R, t = fx.POSES["rotated"]
    for f, dz in enumerate((0.0, -0.02, -0.1)):
        eng.bind_points(fx.cloud(C, N, f, dz=dz))          # replicated cloud
        sm.update(R, t, 1.0, 1.0)
        for _ in range(6):
            eng.update_time()
    full_h = sm.gather("elevation")                        # collective: the whole plane on every rank
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), r0=eng.r0, r1=eng.r1, emap=eng.om.elevation_map[:, eng.r0:eng.r1],
             normal=eng.om.normal_map[:, eng.r0:eng.r1], add_err=np.float32(eng.om.additive_mean_error), full_h=full_h)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,cfg_name,C,N", [(2, "yaml", 130, 20000), (3, "default", 98, 12000), (2, "yaml_norays", 130, 20000)])
def test_strips_equal_single_process(world, cfg_name, C, N, weights):
    import torch.multiprocessing as mp
    import _fixtures as fx
    from oracle import emap_oracle as eo
    outdir = tempfile.mkdtemp()
    mp.spawn(_worker, args=(world, _free_port(), outdir, cfg_name, C, N), nprocs=world, join=True)
    cfg = dict(eo.YAML if cfg_name.startswith("yaml") else eo.DEFAULTS)
    if cfg_name.endswith("norays"):
        cfg["enable_visibility_cleanup"] = False
    eo.lib().eo_set_strip(0, 1 << 30)
    ref = eo.OracleMap(eo.make_params(cfg, cell_n=C, weights=weights))
    R, t = fx.POSES["rotated"]
    for f, dz in enumerate((0.0, -0.02, -0.1)):
        ref.update_map_with_kernel(fx.cloud(C, N, f, dz=dz), R, t, 1.0, 1.0)
        for _ in range(6):
            ref.update_time()
    covered = np.zeros(C, bool)
    for r in range(world):
        g = np.load(os.path.join(outdir, "rank%d.npz" % r))
        r0, r1 = int(g["r0"]), int(g["r1"])
        covered[r0:r1] = True
        # integer / fixed-point accumulators on both sides: the strips reproduce the single map bit for bit
        assert g["emap"].tobytes() == ref.elevation_map[:, r0:r1].tobytes(), "rank %d planes" % r
        assert g["normal"].tobytes() == ref.normal_map[:, r0:r1].tobytes(), "rank %d normals" % r
        assert float(g["add_err"]) == float(ref.additive_mean_error)
        assert g["full_h"].tobytes() == ref.elevation_map[0].tobytes(), "rank %d: gathered plane" % r
    assert covered.all()


def test_strip_layout():
    from elevation_mapping_cupy_amd.sharded import halo_rows_needed, strip_rows
    for C in (202, 1024, 8192, 130):
        for G in (1, 2, 3, 4, 8):
            rows = [strip_rows(C, G, g) for g in range(G)]
            assert rows[0][0] == 0 and rows[-1][1] == C
            assert all(rows[i][1] == rows[i + 1][0] for i in range(G - 1))
    assert halo_rows_needed(3, 1) == 0 and halo_rows_needed(3, 4) == 7
