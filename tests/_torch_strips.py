"""Test infrastructure (not product): the torch.distributed side of the row-strip tests.

``TorchComm`` -- the two collectives of ``ShardedElevationMap._update`` on top of torch.distributed (gloo on CPU / for several ranks on
one GPU, nccl = RCCL where every rank has a GPU); ``TorchStripEngine`` -- ``HipStripEngine`` plus the exchange buffers (torch tensors on
a torch stream) that the stage-by-stage orchestration hands to such a communicator; ``bench_main`` -- the fallback of ``bench.py
--gpus N`` on boxes where the library's own RCCL path cannot run (more ranks than GPUs).  The product package imports none of this:
its multi-GPU frame is ``emap_update_sharded`` (RCCL issued by the C library)."""
from __future__ import annotations

import ctypes as ct
import json
import os
import time

import numpy as np

from elevation_mapping_cupy_amd.sharded import (HipStripEngine, NativeComm, ShardedElevationMap, frame_marches_by_ray, halo_rows_needed,
                                                ray_balanced_weights, strip_rows, strip_stage_bytes)


class TorchComm:
    """The two collectives of the path on top of torch.distributed (nccl=RCCL on GPU, gloo on CPU)."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device  # torch.device or None (CPU)
        # RCCL orders its work after the current HIP stream by itself; gloo with device tensors (test setups with
        # several ranks on one GPU) copies through the host without looking at our stream: drain it first
        self.host_sync = device is not None and "nccl" not in str(dist.get_backend())

    def _drain(self):
        if self.host_sync:
            self.torch.cuda.current_stream(self.device).synchronize()

    def all_reduce_sum_(self, tensor):
        """in-place sum of a small float64 tensor (2 elements: err_sum, err_cnt)."""
        if self.world > 1:
            self._drain()
            self.dist.all_reduce(tensor, op=self.dist.ReduceOp.SUM)
        return tensor

    def exchange_start(self, send_lo, send_hi, recv_lo, recv_hi):
        """neighbour exchange on the strip RING (strips are physical row ranges of a circular map): send_lo -> rank-1 (arrives in
        its recv_hi), send_hi -> rank+1 (arrives in its recv_lo), modulo world.  The posting order (sends low, high; receives
        upper, lower) keeps the pairs apart when both neighbours are the same rank.  Returns the in-flight requests."""
        dist = self.dist
        if self.world == 1:
            return []
        self._drain()
        prev, nxt = (self.rank - 1) % self.world, (self.rank + 1) % self.world
        ops = [dist.P2POp(dist.isend, send_lo, prev), dist.P2POp(dist.isend, send_hi, nxt),
               dist.P2POp(dist.irecv, recv_hi, nxt), dist.P2POp(dist.irecv, recv_lo, prev)]
        return dist.batch_isend_irecv(ops)

    def exchange_wait(self, works):
        for w in works:
            w.wait()
        if self.host_sync:
            self.torch.cuda.synchronize(self.device)

    def all_gather_object(self, obj):
        if self.world == 1:
            return [obj]
        self._drain()
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max_float(self, x):
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.device)
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


class TorchStripEngine(HipStripEngine):
    """HipStripEngine + what a frame driven stage by stage from Python needs: a torch stream the kernels and the collectives are
    ordered on, and the exchange buffers as torch tensors."""

    def __init__(self, param, rank, world, device_index, torch_device, row_weights=None):
        import torch
        self.torch = torch
        self.torch_device = torch_device
        # one dedicated torch stream per strip: the HIP kernels (C ABI) and the collectives are ordered on it
        self.stream = torch.cuda.Stream(device=torch_device)
        super().__init__(param, rank, world, device_index, torch_device, row_weights, stream=self.stream.cuda_stream)
        n = max(1, self.halo * self.C * 4)          # emap_halo_pack: the 16-byte cold half cells of the boundary rows
        with torch.cuda.stream(self.stream):
            mk = lambda: torch.zeros(n, dtype=torch.float32, device=torch_device)  # noqa: E731
            self.send = [mk(), mk()]
            self.recv = [mk(), mk()]
            self.sums = torch.zeros(2, dtype=torch.float64, device=torch_device)
        self.stream.synchronize()

    def stream_ctx(self):
        return self.torch.cuda.stream(self.stream)

    def local_sums(self):
        self._chk(self.lib.emap_drift_sums_to_device(self.ctx, ct.c_void_p(self.sums.data_ptr())))
        return self.sums

    def gate(self, pn, on, totals):
        self._chk(self.lib.emap_set_drift_inputs_device(self.ctx, ct.c_double(pn), ct.c_double(on), ct.c_void_p(totals.data_ptr())))

    def halo_pack(self):
        for side in (0, 1):
            self._chk(self.lib.emap_halo_pack(self.ctx, side, ct.c_void_p(self.send[side].data_ptr())))
        return self.send[0], self.send[1], self.recv[0], self.recv[1]

    def halo_unpack(self, have_lo, have_hi):
        if have_lo:
            self._chk(self.lib.emap_halo_unpack(self.ctx, 0, ct.c_void_p(self.recv[0].data_ptr())))
        if have_hi:
            self._chk(self.lib.emap_halo_unpack(self.ctx, 1, ct.c_void_p(self.recv[1].data_ptr())))

    def normal_halo_pack(self):
        if not hasattr(self, "nsend"):
            n = max(1, 3 * self.halo * self.C)
            with self.torch.cuda.stream(self.stream):
                mk = lambda: self.torch.zeros(n, dtype=self.torch.float32, device=self.torch_device)  # noqa: E731
                self.nsend, self.nrecv = [mk(), mk()], [mk(), mk()]
        for side in (0, 1):
            self._chk(self.lib.emap_normal_halo_pack(self.ctx, side, ct.c_void_p(self.nsend[side].data_ptr())))
        return self.nsend[0], self.nsend[1], self.nrecv[0], self.nrecv[1]

    def normal_halo_unpack(self):
        for side in (0, 1):
            self._chk(self.lib.emap_normal_halo_unpack(self.ctx, side, ct.c_void_p(self.nrecv[side].data_ptr())))


def bench_main(a, rank, world, local_rank):
    """``bench.py --gpus N`` where the library's own RCCL strips cannot run (more ranks than GPUs): row strips of the SAME workload as
    N=1 (strong scaling) with the frame driven stage by stage through torch.distributed."""
    import torch
    import torch.distributed as dist
    from elevation_mapping_cupy_amd.configs import CORE_PARAM_YAML, parameter_from

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import sys
    import _fixtures as fx

    # RCCL prints its version banner on stdout through C stdio: keep fd 1 pointed at stderr until the JSON line is due
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    n_dev = max(1, torch.cuda.device_count())
    oversubscribed = world > n_dev          # more ranks than GPUs (single-GPU boxes): ranks share devices, RCCL refuses that => gloo
    local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # CPU tensors (bootstrap, timing reductions) go through gloo; the nccl backend is only instantiated if the
        # torch-driven fallback below has to move device tensors.  Single node: keep gloo on the loopback interface (the
        # container hostname may not resolve); if gloo cannot come up at all, everything runs over nccl.
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")       # RCCL's bootstrap sockets too (data moves over xGMI / shared memory)
        try:
            dist.init_process_group(backend="gloo" if oversubscribed else "cpu:gloo,cuda:nccl", rank=rank, world_size=world)
        except Exception as ex:  # noqa: BLE001
            print("[rank %d] gloo bootstrap unavailable (%s); using nccl only" % (rank, ex), file=sys.stderr)
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
    cpu_ok = "gloo" in str(dist.get_backend())
    cfg = dict(CORE_PARAM_YAML)
    multimodal = a.workload == "cfg5"
    if a.workload in ("cfg2", "cfg5"):
        cfg.update(enable_visibility_cleanup=False, enable_overlap_clearance=False)
    C, N = a.cell_n, a.points
    if multimodal and C > 2049:
        a.mode = "fp32"
    w = np.load(os.path.join(ROOT, "tests", "golden", "weights.npz"))
    weights = {k: w[k] for k in ("w1", "w2", "w3", "w_out")}
    par = parameter_from(cfg, C, a.mode, weights, device=local_rank)
    # frames with the visibility pass: strips of equal ray work (thin around the sensor) instead of equal height
    row_w = None
    # (a frame that marches its rays BY RAY -- emap_set_ray_mode -- wants equal heights; this engine drives the frame stage by stage
    # from Python, i.e. always by row: frame_marches_by_ray(..., comm_kind="torch") is False whatever the map size)
    if cfg["enable_visibility_cleanup"] and world > 1 and not frame_marches_by_ray(C, N, world, "torch") and os.environ.get("EMAP_STRIPS", "balanced") == "balanced":
        row_w = ray_balanced_weights(C, float(cfg["resolution"]), float(cfg["max_ray_length"]), halo_rows_needed(cfg["dilation_size"], world), world)
    eng = TorchStripEngine(par, rank, world, local_rank, dev, row_w)
    comm, comm_kind = None, ("torch" if oversubscribed else os.environ.get("EMAP_COMM", "native"))

    def all_agree(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=None if cpu_ok else dev)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)      # CPU tensor: gloo
        return int(flag.item()) == 1

    if comm_kind == "native":
        # every step that can fail on one rank only is followed by an agreement, so that no rank is left alone in a collective
        try:
            from elevation_mapping_cupy_amd.sharded import rccl_library_path
            payload = [None]
            if rank == 0:
                uid0 = (ct.c_uint8 * 128)()
                payload[0] = bytes(uid0) if eng.lib.emap_comm_unique_id(rccl_library_path().encode(), uid0) == 0 else None
            if world > 1:
                dist.broadcast_object_list(payload, src=0)
            if payload[0] is None:
                raise RuntimeError("rank 0 could not create the RCCL unique id")
            comm = NativeComm(eng, rank=rank, world=world, bootstrap=False, uid=payload[0])
        except Exception as ex:  # noqa: BLE001
            print("[rank %d] native RCCL path unavailable (%s)" % (rank, ex), file=sys.stderr)
            comm = None
        if all_agree(comm is not None):
            try:
                comm.selftest()
                ok = True
            except Exception as ex:  # noqa: BLE001
                print("[rank %d] RCCL self-test failed (%s)" % (rank, ex), file=sys.stderr)
                ok = False
            if not all_agree(ok):
                comm = None
        else:
            comm = None
        if comm is None and rank == 0:
            print("falling back to torch.distributed collectives", file=sys.stderr)
    if comm is None:
        comm_kind = "torch"
        comm = TorchComm(dev)
    sm = ShardedElevationMap(eng, comm, cfg["enable_visibility_cleanup"], cfg["enable_overlap_clearance"])

    NCLOUD = 2 if multimodal else 5
    channels, stride = None, 3
    if multimodal:                      # rgb (packed 24 bit) + 3 averaged semantic channels, as bench.py builds them at N = 1
        channels, stride = ["x", "y", "z", "rgb", "sem0", "sem1", "sem2"], 7
        par.pointcloud_channel_fusions = {"rgb": "color", "default": "average"}
        host = []
        for s_ in range(NCLOUD):
            p_ = fx.cloud(C, N, s_, dz=-0.02 * s_, extra=4)
            rng = np.random.default_rng(100 + s_)
            p_[:, 3] = rng.integers(0, 1 << 24, N, dtype=np.uint32).view(np.float32)
            p_[:, 4:7] = rng.uniform(0, 1, (N, 3)).astype(np.float32)
            host.append(p_)
        clouds = [torch.from_numpy(p_).to(dev) for p_ in host]
    else:
        clouds = [torch.from_numpy(fx.cloud(C, N, s, dz=(0.0 if s == 0 else -0.02 * s))).to(dev) for s in range(NCLOUD)]
    R = np.eye(3, dtype=np.float32).ravel().copy()
    t = np.array([0, 0, 1], np.float32)

    def frame(i):
        cl = clouds[i % NCLOUD]
        eng.bind_points_device(cl.data_ptr(), N, stride)
        sm.update(R, t, 1.0, 1.0, channels)

    for i in range(3):
        frame(i)
        for _ in range(4):
            eng.update_time()
    eng.update_variance()
    for i in range(a.warmup):
        frame(i)
    torch.cuda.synchronize(); comm.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        frame(i)
    torch.cuda.synchronize()
    wall_local = time.perf_counter() - t0
    comm.barrier(); torch.cuda.synchronize()
    wall = comm.max_float(max(wall_local, 0.0))
    # per-stage device time of THIS rank's strip (hipEvents on the strip's stream) -> roofline of its dominant kernel.
    # Algorithmic bytes of a strip: every rank reads the whole replicated cloud, but sorts / fuses only the points of its
    # rows (N / world for uniform clouds) and streams only its L / world cells.
    roof = None
    if isinstance(comm, NativeComm):
        from elevation_mapping_cupy_amd._lib import STAGES
        eng.lib.emap_enable_stage_timing(eng.ctx, 1)
        reps, acc = min(a.steps, 20), np.zeros(10)
        for i in range(reps):
            frame(i)
            ms10 = (ct.c_float * 10)()
            eng.lib.emap_get_stage_times(eng.ctx, ms10)
            acc += np.array(list(ms10))
        eng.lib.emap_enable_stage_timing(eng.ctx, 0)
        torch.cuda.synchronize(); comm.barrier()
        stage_ms = dict(zip(STAGES, (acc / reps).tolist()))
        strip_bytes = strip_stage_bytes(N, C * C, world, full_sort=bool(cfg["enable_visibility_cleanup"]))
        empty = []                       # spacing of an event pair with nothing in between (bench.py does the same calibration)
        for _ in range(50):
            e_ms = ct.c_float(0)
            eng.lib.emap_timer_begin(eng.ctx); eng.lib.emap_timer_end(eng.ctx, ct.byref(e_ms)); empty.append(e_ms.value)
        ev_overhead = float(np.median(empty))
        kernels = {k: v for k, v in stage_ms.items() if strip_bytes[k] > 0}
        dom = max(kernels, key=kernels.get)
        dom_ms = max(stage_ms[dom] - ev_overhead, 1e-6)
        achieved = strip_bytes[dom] / (dom_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4),
                "traffic": None, "algorithmic_bytes": int(strip_bytes[dom]), "kernel_ms": round(dom_ms, 5), "event_pair_overhead_ms": round(ev_overhead, 5), "rank": 0,
                "stage_ms": {k: round(v, 5) for k, v in stage_ms.items()},
                "note": "rank 0's strip; 'gate' includes the all-reduce, 'post' the halo exchange overlapped with the interior stencils"}
    if rank == 0:
        out = {
            "metric": "Mpoints/s fused (map-update p50 latency in config)", "value": round(N * a.steps / wall / 1e6, 2),
            "unit": "Mpoints/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(wall * 1e3 / a.steps, 5), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %dx%d map in %d row strips, %d uniform-random points/frame replicated to every rank, "
                                   "core_param.yaml values" % (a.workload, C, C, world, N),
                       "index_mode": a.mode, "halo_rows": eng.halo, "parallelism": "row-strips x%d" % world, "ranks": world,
                       "physical_devices": min(n_dev, world), "oversubscribed": bool(oversubscribed),
                       "strip_heights": "equal ray work (thin around the sensor)" if row_w is not None else "equal",
                       "collectives": "all-reduce(2 x f64) + neighbour halo send/recv per frame (RCCL, %s)" %
                                      ("issued by the C library, halo exchange in place on a second stream" if comm_kind == "native"
                                       else "driven through torch.distributed")},
            "roofline": roof, "cpu_baseline": None,
        }
    comm.barrier()
    if isinstance(comm, NativeComm):
        eng.lib.emap_comm_destroy(eng.ctx)
    dist.destroy_process_group()
    try:
        ct.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)
