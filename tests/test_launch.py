"""CPU: the rank launcher and the file rendezvous behind ``bench.py --gpus N`` (elevation_mapping_cupy_amd/launch.py).  The GPU
work itself is skipped (``--dry-run``); what runs is the process model of the multi-GPU bench: N OS processes with RANK /
WORLD_SIZE, the shared rendezvous directory, the agreement protocol, rank 0's single JSON line relayed by the launcher."""
import json
import multiprocessing as mp
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None)
    if env:
        e.update(env)
    return subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=e)


def test_self_launch_prints_one_json_line_with_the_ranks_that_ran():
    out = _run([sys.executable, "bench.py", "--gpus", "3", "--dry-run"])
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["agreed"] is True and [r["rank"] for r in d["ranks"]] == [0, 1, 2]
    assert len({r["pid"] for r in d["ranks"]}) == 3                      # three OS processes
    # what the N > 1 line measures: `value` on BASELINE configs[1], and -- what north_star's scaling target names -- configs[4] (8192^2
    # multi-modal, 16 M points) and configs[3] (4096^2, rays) as sharded sub-measurements, each with a same-box N = 1 run beside it
    plan = d["plan"]
    assert plan["value"] == {"workload": "cfg2", "cell_n": 1024, "points": 1000000, "steps": 50}
    assert plan["config"]["cfg5"] == {"workload": "cfg5", "cell_n": 8192, "points": 16000000, "steps": 10, "n1": True}
    assert plan["config"]["cfg4"] == {"workload": "cfg4", "cell_n": 4096, "points": 4000000, "steps": 10, "n1": True}


def test_a_single_workload_line_plans_no_sub_measurements():
    out = _run([sys.executable, "bench.py", "--gpus", "2", "--dry-run", "--workload", "cfg5"])
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip()][0])
    assert d["plan"]["value"]["workload"] == "cfg5" and d["plan"]["value"]["cell_n"] == 8192 and d["plan"]["config"] == {}


def test_ranks_started_by_an_external_launcher_find_each_other():
    """the driver's launch line (python -m torch.distributed.run ... bench.py --gpus N): WORLD_SIZE is set, bench.py must not spawn"""
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29417", "bench.py", "--gpus", "2", "--dry-run"])
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def _rdv_worker(d, rank, world, q):
    sys.path.insert(0, ROOT)
    from elevation_mapping_cupy_amd.launch import FileRendezvous
    r = FileRendezvous(d, rank, world, timeout=30)
    uid = r.broadcast("uid", bytes(range(128)) if rank == 0 else None)
    ok_all = r.agree("step1", True)
    ok_one_failed = r.agree("step2", rank != 1)                          # rank 1 reports a failure: nobody may proceed
    got = r.gather_json("vals", {"rank": rank, "x": rank * rank})
    r.barrier("end")
    r.finish()
    q.put((rank, uid == bytes(range(128)), ok_all, ok_one_failed, [g["x"] for g in got]))


def test_file_rendezvous_broadcast_agree_gather():
    d = tempfile.mkdtemp(prefix="emap_rdv_test_")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_rdv_worker, args=(d, r, 3, q)) for r in range(3)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=60) for _ in ps)
    for p in ps:
        p.join(timeout=30)
    assert [r[0] for r in res] == [0, 1, 2]
    for _, uid_ok, ok_all, ok_failed, xs in res:
        assert uid_ok and ok_all is True and ok_failed is False and xs == [0, 1, 4]
    assert not os.path.exists(d)                                         # rank 0 removed the directory after everyone left
