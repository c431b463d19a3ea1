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
* the two exchange steps are issued by the C library itself over RCCL (``NativeComm`` -> ``emap_comm_init`` /
  ``emap_update_sharded``): nothing syncs with the host inside a frame, no second communication backend in the product.

The stage ORDER of a sharded frame is also written out engine- and communicator-agnostically (``ShardedElevationMap._update``: what
``emap_update_sharded`` does, as a protocol over ``count / local_sums / gate / fuse / ... / halo_pack / post``) so that the
world_size-2 / 3 gloo tests can drive it on CPU with a test engine and a test communicator (tests/_torch_strips.py holds the
torch.distributed communicator and the engine with exchange buffers; nothing in this package imports torch).
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
    stage-by-stage orchestration (ShardedElevationMap._update) always marches by row.  Strips of equal RAY work (ray_balanced_weights) are for
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


def rccl_library_path():
    """RCCL build that matches the HIP runtime of this process: a process that has PyTorch loaded (its wheels bundle their own ROCm
    libraries and export them globally) must take that librccl.so; otherwise the system ROCm's.  (Looks at sys.modules only.)"""
    env = os.environ.get("EMAP_RCCL_LIB")
    if env:
        return env
    import sys
    loaded = sys.modules.get("torch")
    if loaded is not None and getattr(loaded, "__file__", None):
        cand = os.path.join(os.path.dirname(loaded.__file__), "lib", "librccl.so")
        if os.path.exists(cand):
            return cand
    for cand in ("/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"):
        if os.path.exists(cand):
            return cand
    return "librccl.so.1"


class NativeComm:
    """Both exchange steps issued by the C library itself (``emap_comm_init`` / ``emap_update_sharded``): RCCL resolved with
    dlopen, all-reduce on the strip's stream, in-place halo send/recv on a second stream.  The one out-of-band step -- handing rank
    0's 128-byte ncclUniqueId to every rank -- goes through ``uid`` (the caller distributed it) or a ``launch.FileRendezvous``
    (``rdv``; default: the one the launcher's environment names); barriers and reductions afterwards go through RCCL itself."""

    def __init__(self, engine, rank=None, world=None, bootstrap=True, uid=None, rccl_path=None, rdv=None):
        from ._lib import EmapError
        self.e = engine
        self.rank = int(os.environ.get("RANK", 0) if rank is None else rank)
        self.world = int(os.environ.get("WORLD_SIZE", 1) if world is None else world)
        path = (rccl_path or rccl_library_path()).encode()
        if uid is not None:              # the caller distributed the id itself
            uid = (ct.c_uint8 * 128).from_buffer_copy(bytes(uid))
        else:
            uid = (ct.c_uint8 * 128)()
            ok = 1
            if self.rank == 0:
                ok = 1 if engine.lib.emap_comm_unique_id(path, uid) == 0 else 0
            if self.world > 1:
                if not bootstrap and rdv is None:
                    raise EmapError("several ranks need the unique id handed over (uid=) or a rendezvous (rdv=)")
                from .launch import FileRendezvous
                rdv = rdv or FileRendezvous.from_env(self.rank, self.world)
                # agree on success before the collective init (a rank that cannot load RCCL must not leave the others waiting)
                payload = rdv.broadcast("rccl_uid", (bytes(uid) if ok else b"") if self.rank == 0 else None)
                if len(payload) != 128:
                    raise EmapError("rank 0 could not create the RCCL unique id")
                uid = (ct.c_uint8 * 128).from_buffer_copy(payload)
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

    def _reduce(self, x, op):
        v = (ct.c_double * 1)(float(x))
        self.e._chk(self.e.lib.emap_comm_allreduce_host(self.e.ctx, v, 1, int(op)))
        return float(v[0])

    def barrier(self):
        """behind all work enqueued on the strip's stream, on every rank (an all-reduce through the communicator itself)"""
        if self.world > 1:
            self._reduce(0.0, 0)

    def max_float(self, x):
        return float(x) if self.world == 1 else self._reduce(x, 1)


# ---------------------------------------------------------------------------------------------------------------
class HipStripEngine:
    """One strip on one MI355X: thin adapter from the sharding protocol to the C ABI.  Whole frames run inside the library
    (``update_native`` -> ``emap_update_sharded`` over a ``NativeComm``); the single stages are exposed for orchestrations that drive
    a frame stage by stage (``ShardedElevationMap._update`` with an engine that also owns exchange buffers: the test engines)."""

    def __init__(self, param, rank, world, device_index, torch_device=None, row_weights=None, stream=None):
        from .elevation_mapping import ElevationMap
        C = int(param.cell_n)
        r0, r1 = strip_rows(C, world, rank, row_weights)
        self.halo = halo_rows_needed(param.dilation_size, world)
        if world > 1 and (r1 - r0) < self.halo:
            raise ValueError("strip of %d rows is thinner than the %d-row halo" % (r1 - r0, self.halo))
        param.device = device_index
        self.map = ElevationMap(param, strip=(r0, r1 - r0, self.halo), stream=stream)      # stream None: the context's own
        self.lib, self.ctx = self.map._lib, self.map._ctx
        self.C, self.rows = C, r1 - r0

    def _chk(self, rc):
        self.map._chk(rc)

    def bind_points_device(self, ptr, n, stride):
        self.map.bind_points_device(ptr, n, stride)

    def bind_points(self, points):
        self.map.bind_points(points)

    def count(self, R, t):
        self.map.stage("count", R, t)

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

    def post(self, part=0):
        """dilation + traversability + normals; part 1 = tiles independent of the halo, 2 = boundary tiles, 0 = all"""
        self._chk(self.lib.emap_post_part(self.ctx, int(part)))

    def normal_row_lag(self):
        """rows by which the (never shifted) normal planes lag the cells since the last row shift"""
        lag = ct.c_int32(0)
        self._chk(self.lib.emap_normal_row_lag(self.ctx, ct.byref(lag)))
        return lag.value

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

    def semantic_declare(self, channels):
        """the strip's RGB / semantic fusion rides inside the next frame (SemanticMap.declare_frame); returns the jobs to run after it"""
        return self.map.semantic_map.declare_frame(self.map, list(channels))

    def semantic_finish(self, jobs, R, t):
        R = np.ascontiguousarray(np.asarray(R, np.float32).reshape(9))
        t = np.ascontiguousarray(np.asarray(t, np.float32).reshape(3))
        self.map.semantic_map.finish_frame(self.map, jobs, R, t)

    def semantic_update(self, channels, R, t):
        """RGB / semantic fusion of the bound cloud's extra channels into this strip's layers AFTER a frame: per cell, no exchange step"""
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
        if isinstance(c, NativeComm):
            jobs = e.semantic_declare(extra) if extra else []      # the fusion rides inside the library's frame (emap_frame_semantics)
            e.update_native(R, t, position_noise, orientation_noise)
            if jobs:
                e.semantic_finish(jobs, R, t)
            return
        if extra:
            e.semantic_prepare(extra)
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


