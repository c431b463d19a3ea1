"""Row-strip sharding of the elevation map over the GPUs of one node (one process per GPU).

New design -- the reference is single-GPU (SURVEY §2.1 "Parallelism strategies: none").  Cell index = C*ix + iy
(reference custom_kernels.py:45-49), so rank g owns the contiguous rows ``[g*C//G, (g+1)*C//G)`` of every array.

Per frame (same stage order as reference update_map_with_kernel, EM/elevation_mapping.py:316-391):

    count (local rows) -> ALL-REDUCE(sum) of (err_sum, err_cnt) -> gate on the totals -> fuse -> [commit -> rays]
    -> average -> overlap clearance -> HALO EXCHANGE of `halo` rows of cells with the strip neighbours
    -> dilation (owned rows +-3) -> traversability + normals

* every rank is handed the same cloud and there is no exchange for fusion: a rank either binds all of it and keeps the points whose
  row it owns (in-kernel), or -- round 5, wherever the frame allows it -- converts / uploads only the points that can land in its
  rows (``ShardedElevationMap.input_pointcloud`` -> ``emap_upload_points_strip``: 1 / G of the cloud per rank);
* rays marched BY ROW need no communication: every rank marches every ray and acts only on its own rows; the library's own frame
  (``emap_update_sharded``) marches them BY RAY over an all-reduced window from 2048^2 cells on;
* the two exchange steps go through ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on the GPU box,
  ``gloo`` in the CPU tests); buffers are plain device pointers on the C-ABI side (``emap_halo_pack/unpack``,
  ``emap_drift_sums_to_device``), so nothing syncs with the host inside a frame.

The orchestration is engine-agnostic (``StripEngine`` protocol) so that the world_size-2 gloo tests can drive it on
CPU with a test engine; the product engine is ``HipStripEngine`` (libemap_hip.so).
"""
from __future__ import annotations

import contextlib
import ctypes as ct
import json
import os
import time

import numpy as np


def strip_rows(cell_n: int, world: int, rank: int, weights=None):
    """Rows [begin, end) owned by ``rank``.  ``weights`` (one non-negative number per row, see ``ray_balanced_weights``) splits
    the rows into contiguous strips of equal cumulative weight instead of equal height."""
    if weights is None:
        return (rank * cell_n) // world, ((rank + 1) * cell_n) // world
    w = np.asarray(weights, np.float64)
    assert w.shape == (cell_n,) and (w >= 0).all() and w.sum() > 0
    cum = np.concatenate([[0.0], np.cumsum(w)]) / w.sum()
    cuts = [int(np.searchsorted(cum, g / world, side="left")) for g in range(world + 1)]
    cuts[0], cuts[-1] = 0, cell_n
    for g in range(1, world + 1):                     # strictly increasing, so that no strip is empty
        cuts[g] = max(cuts[g], cuts[g - 1] + 1)
    for g in range(world - 1, -1, -1):
        cuts[g] = min(cuts[g], cuts[g + 1] - 1)
    return cuts[rank], cuts[rank + 1]


def ray_balanced_weights(cell_n: int, resolution: float, max_ray_length: float, min_rows: int, world: int, cell_weight: float = 0.08):
    """Row weights for frames WITH the visibility pass and a sensor near the map centre (robot-centric maps): every ray starts at the
    sensor, so only the rows within ``max_ray_length`` of the centre carry ray work and -- a wave runs as long as its longest ray --
    that work is proportional to the height of a strip inside this band (tools/exp_strip_rays.py: 0.79 ms for each of the two centre
    strips of an 8-way uniform split of the 1024^2 / 1 M workload, 0.02 ms for the outer ones).  ``cell_weight`` is the per-row cost of
    the per-cell stages relative to a row inside the band.  Strips never get thinner than ``min_rows`` (the halo)."""
    rows = np.arange(cell_n) + 0.5 - cell_n / 2.0
    w = np.where(np.abs(rows) * resolution <= max_ray_length, 1.0, 0.0) + cell_weight
    if (w.sum() / world) / w.max() < max(1, min_rows):     # the thinnest strip would be thinner than its halo: keep equal heights
        return None
    return w


def frame_marches_by_ray(cell_n: int, n_points: int, world: int, comm_kind: str = "native", ray_mode: int = 0, scatter: str = "auto") -> bool:
    """Will a sharded frame WITH the visibility pass march its rays by ray (csrc/emap_api.hip: rays_by_ray -- the same predicate, from
    the values every rank shares)?  Only the library's own frame (``emap_update_sharded`` over the native RCCL communicator) can; the
    torch-driven fallback and the staged strip engine always march by row.  Strips of equal RAY work (ray_balanced_weights) are for
    frames that march by row; a by-ray frame wants equal heights (the decision must not hang on cell_n alone)."""
    binned = scatter == "binned" or (scatter == "auto" and n_points >= 131072)
    return world > 1 and comm_kind == "native" and ray_mode != 1 and binned and (ray_mode == 2 or cell_n >= 2048)


def strip_stage_bytes(n_points: int, n_cells: int, world: int, full_sort: bool = False, bucketed: bool = False):
    """Algorithmic bytes of the timed stages of ONE strip's frame (the kernels as they are; bench.py: STAGE_BYTES holds the
    single-context table): N points in the frame's cloud, L cells in the whole map.  The point passes of a strip stream the
    replicated cloud once (12 B of xyz per point; `bucketed`: only the strip's share was uploaded) and keep a 16-byte staging record
    per OWNED point, which the scatter pass permutes; a frame whose rays march by row sorts every valid point (`full_sort`: the
    single-context point passes).  The tile and stencil kernels only see the strip's points and cells."""
    N, L = float(n_points), float(n_cells)
    Nw, Lw = N / world, L / world
    if full_sort:
        hist, scatter = 12 * N, 12 * N + 16 * N
    else:
        hist, scatter = 12 * (Nw if bucketed else N) + 16 * Nw, 16 * Nw + 16 * Nw
    return {"hist": hist, "scan": 0, "scatter": scatter, "gate": 16 * Nw + 16 * Lw, "fuse": 16 * Nw + 32 * Lw, "commit": 104 * Lw,
            "rays": 12 * N + 48 * Lw, "average": 120 * Lw, "overlap": 0, "post": 40 * Lw}


def halo_rows_needed(dilation_size: int, world: int) -> int:
    """dilation radius d, +3 rows for the traversability stencil computed from the dilated plane, +1 for the
    reference's flat-index wrap into the adjacent row (custom_kernels.py:403-407)."""
    return 0 if world == 1 else int(dilation_size) + 4


# ---------------------------------------------------------------------------------------------------------------
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


def rccl_library_path():
    """RCCL build that matches the HIP runtime of this process: PyTorch wheels bundle their own ROCm libraries and export
    them globally, so with torch imported its librccl.so is the consistent choice; otherwise the system ROCm's."""
    env = os.environ.get("EMAP_RCCL_LIB")
    if env:
        return env
    import sys
    if "torch" in sys.modules:
        cand = os.path.join(os.path.dirname(sys.modules["torch"].__file__), "lib", "librccl.so")
        if os.path.exists(cand):
            return cand
    for cand in ("/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"):
        if os.path.exists(cand):
            return cand
    return "librccl.so.1"


class NativeComm:
    """Both exchange steps issued by the C library itself (``emap_comm_init`` / ``emap_update_sharded``): RCCL resolved with
    dlopen, all-reduce on the strip's stream, in-place halo send/recv on a second stream.  torch.distributed (any backend)
    is only the bootstrap channel for the 128-byte ncclUniqueId and the out-of-band barrier / timing reductions."""

    def __init__(self, engine, rank=None, world=None, bootstrap=True, uid=None, rccl_path=None):
        from ._lib import EmapError
        self.e = engine
        if bootstrap:
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            self.rank, self.world = dist.get_rank(), dist.get_world_size()
        else:
            self.torch = self.dist = None
            self.rank, self.world = int(rank or 0), int(world or 1)
        path = (rccl_path or rccl_library_path()).encode()
        ok = 1
        if uid is not None:              # the caller distributed the id itself (bootstrap=False with several ranks)
            uid = (ct.c_uint8 * 128).from_buffer_copy(bytes(uid))
        else:
            uid = (ct.c_uint8 * 128)()
            if self.rank == 0:
                ok = 1 if engine.lib.emap_comm_unique_id(path, uid) == 0 else 0
        if self.world > 1 and bootstrap:
            # agree on success before the collective init (a rank that cannot load RCCL must not leave the others waiting)
            payload = [bytes(uid) if ok else None]
            self.dist.broadcast_object_list(payload, src=0)
            if payload[0] is None:
                raise EmapError("rank 0 could not create the RCCL unique id")
            uid = (ct.c_uint8 * 128).from_buffer_copy(payload[0])
        elif not ok:
            raise EmapError("could not create the RCCL unique id (%s)" % path.decode())
        engine._chk(engine.lib.emap_comm_init(engine.ctx, path, uid, self.rank, self.world))
        self.path = path.decode()

    def selftest(self):
        self.e._chk(self.e.lib.emap_comm_selftest(self.e.ctx))

    def gather_layer(self, plane_id):
        """(cell_n, cell_n) plane of the FULL map on every rank (emap_comm_gather_layer: an exact all-reduce of zero-padded planes)"""
        C = self.e.C
        out = np.empty((C, C), np.float32)
        self.e._chk(self.e.lib.emap_comm_gather_layer(self.e.ctx, int(plane_id), out.ctypes.data_as(ct.POINTER(ct.c_float))))
        return out

    def rccl_ranks(self):
        """size of the live RCCL communicator as RCCL reports it (ncclCommCount)"""
        n = ct.c_int32(0)
        self.e._chk(self.e.lib.emap_comm_count(self.e.ctx, ct.byref(n)))
        return int(n.value)

    def _oob(self, t):
        """tensor for the out-of-band (bootstrap) channel: CPU when gloo is available, else on the strip's device"""
        return t if "gloo" in str(self.dist.get_backend()) else t.to(self.e.torch_device)

    def barrier(self):
        if self.world > 1:      # out-of-band (host) barrier on the bootstrap channel; callers synchronise the device themselves
            self.dist.all_reduce(self._oob(self.torch.zeros(1, dtype=self.torch.int32)))

    def max_float(self, x):
        if self.world == 1:
            return float(x)
        t = self._oob(self.torch.tensor([x], dtype=self.torch.float64))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


# ---------------------------------------------------------------------------------------------------------------
class HipStripEngine:
    """One strip on one MI355X: thin adapter from the sharding protocol to the C ABI."""

    def __init__(self, param, rank, world, device_index, torch_device, row_weights=None):
        import torch
        from .elevation_mapping import ElevationMap
        self.torch = torch
        self.torch_device = torch_device
        C = int(param.cell_n)
        r0, r1 = strip_rows(C, world, rank, row_weights)
        self.halo = halo_rows_needed(param.dilation_size, world)
        if world > 1 and (r1 - r0) < self.halo:
            raise ValueError("strip of %d rows is thinner than the %d-row halo" % (r1 - r0, self.halo))
        param.device = device_index
        # one dedicated torch stream per strip: the HIP kernels (C ABI) and the collectives are ordered on it
        self.stream = torch.cuda.Stream(device=torch_device)
        self.map = ElevationMap(param, strip=(r0, r1 - r0, self.halo), stream=self.stream.cuda_stream)
        self.lib, self.ctx = self.map._lib, self.map._ctx
        self.C, self.rows = C, r1 - r0
        n = max(1, self.halo * C * 4)          # emap_halo_pack: the 16-byte cold half cells of the boundary rows
        with torch.cuda.stream(self.stream):
            mk = lambda: torch.zeros(n, dtype=torch.float32, device=torch_device)  # noqa: E731
            self.send = [mk(), mk()]
            self.recv = [mk(), mk()]
            self.sums = torch.zeros(2, dtype=torch.float64, device=torch_device)
        self.stream.synchronize()

    def stream_ctx(self):
        return self.torch.cuda.stream(self.stream)

    def _chk(self, rc):
        self.map._chk(rc)

    def bind_points_device(self, ptr, n, stride):
        self.map.bind_points_device(ptr, n, stride)

    def bind_points(self, points):
        self.map.bind_points(points)

    def count(self, R, t):
        self.map.stage("count", R, t)

    def local_sums(self):
        self._chk(self.lib.emap_drift_sums_to_device(self.ctx, ct.c_void_p(self.sums.data_ptr())))
        return self.sums

    def gate(self, pn, on, totals):
        self._chk(self.lib.emap_set_drift_inputs_device(self.ctx, ct.c_double(pn), ct.c_double(on), ct.c_void_p(totals.data_ptr())))

    def fuse(self, R, t):
        self.map.stage("fuse", R, t)

    def fuse_average(self, R, t):
        self.map.stage("fuse_average", R, t)

    def commit(self):
        self.map.stage("commit")

    def rays(self, R, t):
        self.map.stage("rays", R, t)

    def average(self):
        self.map.stage("average")

    def overlap(self, tz):
        self.map.stage("overlap", t=tz)

    def halo_pack(self):
        for side in (0, 1):
            self._chk(self.lib.emap_halo_pack(self.ctx, side, ct.c_void_p(self.send[side].data_ptr())))
        return self.send[0], self.send[1], self.recv[0], self.recv[1]

    def halo_unpack(self, have_lo, have_hi):
        if have_lo:
            self._chk(self.lib.emap_halo_unpack(self.ctx, 0, ct.c_void_p(self.recv[0].data_ptr())))
        if have_hi:
            self._chk(self.lib.emap_halo_unpack(self.ctx, 1, ct.c_void_p(self.recv[1].data_ptr())))

    def post(self, part=0):
        """dilation + traversability + normals; part 1 = tiles independent of the halo, 2 = boundary tiles, 0 = all"""
        self._chk(self.lib.emap_post_part(self.ctx, int(part)))

    def normal_row_lag(self):
        """rows by which the (never shifted) normal planes lag the cells since the last row shift"""
        lag = ct.c_int32(0)
        self._chk(self.lib.emap_normal_row_lag(self.ctx, ct.byref(lag)))
        return lag.value

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

    def move_to(self, position, R):
        """every rank shifts its strip by the same amount: the strip keeps its PHYSICAL rows, the logical rows it holds change"""
        self.map.move_to(position, R)

    def move(self, delta_position):
        self.map.move(delta_position)

    def update_native(self, R, t, position_noise, orientation_noise):
        """whole frame incl. both exchange steps inside the library (needs NativeComm)"""
        R = np.ascontiguousarray(np.asarray(R, np.float32).reshape(9))
        t = np.ascontiguousarray(np.asarray(t, np.float32).reshape(3))
        self._chk(self.lib.emap_update_sharded(self.ctx, R.ctypes.data_as(ct.POINTER(ct.c_float)), t.ctypes.data_as(ct.POINTER(ct.c_float)),
                                               ct.c_double(position_noise), ct.c_double(orientation_noise), None))

    def semantic_prepare(self, channels):
        """create the layers / count plane of the extra cloud channels BEFORE the frame (SemanticMap.prepare)"""
        self.map.semantic_map.prepare(list(channels))

    def semantic_update(self, channels, R, t):
        """RGB / semantic fusion of the bound cloud's extra channels into this strip's layers: per cell, no exchange step"""
        R = np.ascontiguousarray(np.asarray(R, np.float32).reshape(9))
        t = np.ascontiguousarray(np.asarray(t, np.float32).reshape(3))
        self.map.semantic_map.update_layers_pointcloud(self.map, list(channels), R, t)

    def update_time(self):
        self.map.update_time()

    def update_variance(self):
        self.map.update_variance()

    def owned_planes(self):
        return self.map.elevation_map

    def sync(self):
        self.map.sync()


# ---------------------------------------------------------------------------------------------------------------
class ShardedElevationMap:
    """Frame orchestration over row strips; ``engine`` implements the stage protocol for the local strip."""

    def __init__(self, engine, comm, enable_visibility_cleanup, enable_overlap_clearance):
        self.e, self.comm = engine, comm
        self.rays_on, self.overlap_on = bool(enable_visibility_cleanup), bool(enable_overlap_clearance)

    def gather(self, name):
        """one plane of the FULL map, assembled from the strips, on every rank (collective).  ``name``: a core layer name, "normal_x" /
        "normal_y" / "normal_z" or "traversability_input".  Row r of the result is logical map row r, like the single-context map."""
        from ._lib import PLANES
        pid = PLANES[name] if isinstance(name, str) else int(name)
        if hasattr(self.comm, "gather_layer"):
            return self.comm.gather_layer(pid)
        # generic communicators (torch.distributed / gloo): all-gather of (logical begin, rows) through the comm's object channel
        m = self.e.map
        mine = (m.logical_row_begin, m.get_layer_raw(pid))
        parts = self.comm.all_gather_object(mine)
        C = mine[1].shape[1]
        full = np.zeros((C, C), np.float32)
        for b, rows in parts:
            full[(b + np.arange(rows.shape[0])) % C] = rows
        return full

    def move_to(self, position, R):
        self.e.move_to(position, R)

    def move(self, delta_position):
        self.e.move(delta_position)

    def buckets_clouds(self, n_points, channels=None):
        """Can this map's frames run on a cloud BUCKETED per rank (every rank uploads only the points of its rows,
        emap_upload_points_strip)?  Needs the library's own frame (native communicator), no visibility pass that marches by row
        (every valid point marches a ray through every strip then), and no fusion that decodes the global point index."""
        if not isinstance(self.comm, NativeComm) or self.comm.world <= 1:
            return False
        m = self.e.map
        C = int(m.param.cell_n) if hasattr(m, "param") else self.e.C
        by_ray = frame_marches_by_ray(C, int(n_points), self.comm.world, "native", getattr(m, "_ray_mode", 0), getattr(m, "_scatter_mode", "auto"))
        if self.rays_on and not by_ray:
            return False
        if channels is not None and len(channels) > 3 and m.semantic_map is not None:
            _, fusions = m.semantic_map.prepare(list(channels[3:]))
            if any(f in ("pointcloud_class_bayesian", "pointcloud_bayesian_inference", "pointcloud_class_max", "class_bayesian", "bayesian_inference", "class_max") for f in fusions):
                return False
        return True

    def input_pointcloud(self, points, channels, R, t, position_noise, orientation_noise):
        """ElevationMap.input_pointcloud (EM/elevation_mapping.py:434-466) on a sharded map: EVERY rank is handed the same sensor cloud
        (``points``: host (N, 3 + K)); ``t`` is map-centre relative.  Where the frame allows it (buckets_clouds) a rank converts,
        uploads and streams only the points of its rows -- 1 / world of the cloud per rank instead of all of it."""
        pts = np.asarray(points)
        names = list(channels) if channels is not None else None
        if self.buckets_clouds(pts.shape[0], names):
            self.e.map.bind_points(pts, strip_pose=(R, t))
        else:
            self.e.map.bind_points(pts)
        self.update(R, t, position_noise, orientation_noise, names if (names is not None and len(names) > 3) else None)

    def update(self, R, t, position_noise, orientation_noise, channels=None):
        """One frame on the bound cloud (replicated, or bucketed by input_pointcloud); ``t`` is map-centre relative.  ``channels`` (names of ALL cloud columns,
        x, y, z first) additionally fuses the extra columns into the strip's RGB / semantic layers (BASELINE config 5)."""
        e, c = self.e, self.comm
        extra = list(channels[3:]) if channels is not None else None      # x, y, z are not layers (input_pointcloud forwards channels[3:])
        if extra:
            e.semantic_prepare(extra)
        if isinstance(c, NativeComm):
            e.update_native(R, t, position_noise, orientation_noise)
        else:
            ctx = e.stream_ctx() if hasattr(e, "stream_ctx") else contextlib.nullcontext()
            with ctx:
                self._update(R, t, position_noise, orientation_noise)
        if extra:
            e.semantic_update(extra, R, t)

    def _update(self, R, t, position_noise, orientation_noise):
        e, c = self.e, self.comm
        e.count(R, t)
        totals = c.all_reduce_sum_(e.local_sums())            # exchange step 1: 2 scalars
        e.gate(position_noise, orientation_noise, totals)
        if self.rays_on:
            e.fuse(R, t)
            e.commit()
            if c.world > 1 and hasattr(e, "normal_row_lag") and e.normal_row_lag() != 0:
                # a row shift since the last frame: the un-shifted normal planes sit `lag` rows off -- fetch the neighbours' rows.
                # This Python-driven fallback only moves the planes' halo rows; a larger shift needs rows from beyond them, which the
                # library's own frame fetches from whoever owns them (emap_update_sharded: normal_exchange).  Refuse instead of
                # marching the rays over zeros.
                if abs(e.normal_row_lag()) > getattr(e, "halo", 0):
                    from ._lib import EmapError
                    raise EmapError("the map moved %d rows since the normals were written, more than the %d halo rows this fallback exchanges: "
                                    "use the native communicator (emap_comm_init + emap_update_sharded)" % (abs(e.normal_row_lag()), getattr(e, "halo", 0)))
                n_lo, n_hi, q_lo, q_hi = e.normal_halo_pack()
                c.exchange_wait(c.exchange_start(n_lo, n_hi, q_lo, q_hi))
                e.normal_halo_unpack()
            e.rays(R, t)
            e.average()
        else:
            e.fuse_average(R, t)                               # one tile kernel: fuse + commit + average
        if self.overlap_on:
            e.overlap(float(np.float32(np.asarray(t, np.float32).reshape(3)[2])))
        if c.world > 1:                                        # exchange step 2: halo rows of cells ...
            s_lo, s_hi, r_lo, r_hi = e.halo_pack()
            works = c.exchange_start(s_lo, s_hi, r_lo, r_hi)
            e.post(1)                                          # ... overlapped with the stencils of the interior tiles
            c.exchange_wait(works)
            e.halo_unpack(True, True)                          # ring: every strip has both neighbours
            e.post(2)
        else:
            e.post(0)


# ---------------------------------------------------------------------------------------------------------------
def bench_main(a, rank, world, local_rank):
    """``bench.py --gpus N`` under torch.distributed.run: row strips of the SAME workload as N=1 (strong scaling)."""
    import torch
    import torch.distributed as dist
    from .configs import CORE_PARAM_YAML, parameter_from

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
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
    eng = HipStripEngine(par, rank, world, local_rank, dev, row_w)
    comm, comm_kind = None, ("torch" if oversubscribed else os.environ.get("EMAP_COMM", "native"))

    def all_agree(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=None if cpu_ok else dev)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)      # CPU tensor: gloo
        return int(flag.item()) == 1

    if comm_kind == "native":
        # every step that can fail on one rank only is followed by an agreement, so that no rank is left alone in a collective
        try:
            comm = NativeComm(eng)
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
        from ._lib import STAGES
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
