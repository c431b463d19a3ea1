"""One process per GPU without a launcher dependency: rank spawning and a file rendezvous for the row-strip path.

The native multi-GPU path needs exactly one out-of-band exchange -- the 128-byte RCCL unique id that rank 0 creates
(``emap_comm_unique_id``) and every rank hands to ``emap_comm_init`` -- plus agreement points so that no rank is left alone in a
collective when another one failed.  On one node a directory under /tmp is all that takes; everything after the communicator is
up (barriers, timing reductions) goes through RCCL itself (``emap_comm_allreduce_host``).  No torch, no MPI.

``spawn_ranks`` is what ``bench.py --gpus N`` uses when it was not started by an external launcher (``WORLD_SIZE`` unset);
under ``python -m torch.distributed.run`` the ranks already exist and only ``FileRendezvous`` is used (keyed by the launcher's
pid and MASTER_PORT, which all ranks of one job share).
"""
from __future__ import annotations

import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time


class RendezvousError(RuntimeError):
    pass


class FileRendezvous:
    """publish / fetch of small blobs between the ranks of one node through a shared directory (atomic rename, polling)."""

    def __init__(self, directory, rank, world, timeout=120.0):
        self.dir, self.rank, self.world, self.timeout = directory, int(rank), int(world), float(timeout)
        os.makedirs(self.dir, exist_ok=True)

    @classmethod
    def from_env(cls, rank, world, timeout=120.0):
        d = os.environ.get("EMAP_RDV_DIR")
        if not d:     # ranks started by an external launcher: same parent pid + MASTER_PORT on every rank of the job
            d = os.path.join(tempfile.gettempdir(), "emap_rdv_%d_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0")))
        return cls(d, rank, world, timeout)

    def _path(self, name, rank):
        return os.path.join(self.dir, "%s.%d" % (name, rank))

    def publish(self, name, payload: bytes):
        tmp = self._path(name, self.rank) + ".tmp"
        with open(tmp, "wb") as f:
            f.write(payload)
        os.replace(tmp, self._path(name, self.rank))

    def fetch(self, name, rank):
        p, t0 = self._path(name, rank), time.monotonic()
        while not os.path.exists(p):
            if time.monotonic() - t0 > self.timeout:
                raise RendezvousError("rank %d: timed out waiting for '%s' of rank %d in %s" % (self.rank, name, rank, self.dir))
            time.sleep(0.002)
        with open(p, "rb") as f:
            return f.read()

    def broadcast(self, name, payload=None, src=0):
        if self.rank == src:
            self.publish(name, payload)
            return payload
        return self.fetch(name, src)

    def gather_json(self, name, obj):
        """every rank contributes one JSON-able object; every rank gets the list (index = rank)"""
        self.publish(name, json.dumps(obj).encode())
        return [json.loads(self.fetch(name, r).decode()) for r in range(self.world)]

    def agree(self, name, ok):
        """True iff every rank reports success for step `name` (each step name may be used once)"""
        return all(self.gather_json("agree_" + name, bool(ok)))

    def barrier(self, name):
        self.gather_json("barrier_" + name, 1)

    def finish(self):
        """last call of every rank: each rank leaves an exit marker as its final access, rank 0 removes the directory once all are there"""
        self.publish("exit", b"1")
        if self.rank == 0:
            try:
                for r in range(self.world):
                    self.fetch("exit", r)
            finally:
                shutil.rmtree(self.dir, ignore_errors=True)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def device_count():
    """HIP devices visible to this process (0 without a runtime) -- plain ctypes, no torch"""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
    except OSError:
        return 0
    n = ctypes.c_int(0)
    return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0


def spawn_ranks(script, argv, world, extra_env=None, timeout=None):
    """Start `world` copies of `script argv` as ranks 0..world-1 of one node; rank 0's stdout is returned (the JSON line), the
    other ranks' stdout goes to stderr.  Returns (returncode, rank0_stdout)."""
    rdv = tempfile.mkdtemp(prefix="emap_rdv_")
    port = free_port()
    procs = []
    try:
        for r in range(world):
            env = dict(os.environ)
            env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                        "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "EMAP_RDV_DIR": rdv})
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if extra_env:
                env.update(extra_env)
            procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env,
                                          stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=(r == 0)))
        out0 = ""
        deadline = None if timeout is None else time.monotonic() + timeout
        try:
            out0, _ = procs[0].communicate(timeout=timeout)
            rc = procs[0].returncode
            for p in procs[1:]:
                left = None if deadline is None else max(1.0, deadline - time.monotonic())
                rc = p.wait(timeout=left) or rc
        except subprocess.TimeoutExpired:
            rc = 124
        return rc, out0
    finally:
        for p in procs:                       # exact pids we started, never a pattern
            if p.poll() is None:
                p.kill()
        shutil.rmtree(rdv, ignore_errors=True)
