"""GPU: bench.py prints exactly one JSON line with the fields the driver reads (small sizes, seconds)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("workload", ["cfg2", "cfg3"])
def test_bench_json_contract(workload):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--workload", workload,
                          "--cell-n", "202", "--points", "40000", "--cpu-points", "5000"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "stdout must hold ONE JSON line, got: %r" % lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] > 0 and d["unit"] == "Mpoints/s" and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    if workload == "cfg3":          # the visibility march is bound by vector-instruction issue: reported against that bound, the HBM figure beside it
        assert r["bound"] == "valu" and r["kernel"] == "rays" and r["unit"] == "G wave-instructions/s" and abs(r["peak"] - 1228.8) < 0.1 and "traffic" in r
        assert r["frac"] is None or abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3      # (None: no SQ-counter pass of THESE kernel sources at this map size)
        h = r["hbm"]
        assert h["peak"] == 8000.0 and abs(h["frac"] - h["achieved"] / h["peak"]) < 1e-3 and r["ray_samples_per_frame"] > 0
    else:
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["value"] > 0 and c["cores"] >= 1 and c["kind"] == "port" and c["unit"] == "Mpoints/s" and c["sample"]
    if workload == "cfg2":          # the default line also carries the frame WITH the visibility pass, timed in the same process
        c3 = d["config"]["cfg3"]
        assert c3["value"] > 0 and c3["ms_per_step"] > 0 and c3["ray_visits_per_frame"] > 0 and c3["dominant_kernel"] in c3["stage_ms"]


@pytest.mark.parametrize("world,C,N", [(2, 202, 40000), (8, 514, 200000)])
def test_bench_gpus_n_launches_its_own_ranks(world, C, N):
    """`python bench.py --gpus N` without a launcher: N rank processes, one JSON line, n_gpus = the ranks that really ran -- N = 8 is the
    driver's scaling run: rank spawning, the file rendezvous, eight strips with halos and the exchange steps of every frame.  With
    fewer devices than ranks the ranks share them (RCCL refuses that: the stage-by-stage orchestration over gloo carries the frame,
    tests/_torch_strips.py); with N devices the library's own RCCL path runs."""
    env = dict(os.environ, EMAP_BENCH_SUB_SIZES="cfg5:1024:300000,cfg4:1024:300000"); env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "4", "--warmup", "1", "--cell-n", str(C),
                          "--points", str(N), "--cpu-points", "5000", "--no-cpu-baseline"], capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["config"]["ranks"] == world and d["value"] > 0 and d["steps"] == 4
    assert 1 <= d["config"]["physical_devices"] <= world


def test_sharded_default_line_carries_cfg5_and_cfg4_next_to_their_n1_times():
    """The N > 1 line (here: the row-strip path forced onto ONE rank, real RCCL communicator) keeps BASELINE configs[1] as `value` and
    adds what north_star's scaling target names as sharded sub-measurements -- config.cfg5 (multi-modal map, semantic fusion inside
    the timed frame's tile pass; repeated at top level as `scaling_value`) and config.cfg4 (rays + overlap) -- each with the same-box N = 1 time beside it.
    Sub-measurement sizes shrunk through the test hook; the code path is the one `bench.py --gpus 8` runs."""
    env = dict(os.environ, EMAP_BENCH_SUB_SIZES="cfg5:1024:300000,cfg4:1024:300000", EMAP_BENCH_BUCKET="force")      # (the bucketed-cloud code path of N > 1, on one rank)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-sharded", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["rccl_ranks"] == 1 and "cfg2" in d["config"]["workload"]
    for k in ("cfg5", "cfg4"):
        c = d["config"][k]
        assert c["ms_per_step"] > 0 and c["n1_ms_per_step"] > 0 and c["speedup_vs_n1"] > 0 and c["rccl_ranks"] == 1 and c["n_gpus"] == 1
        assert len(c["per_rank_stage_ms"]) == 1 and c["dominant_kernel"] in c["stage_ms_rank0"]
    # the fusion of the extra channels is declared for the frame (emap_frame_semantics) and runs inside its tile pass: no stage of its own
    assert d["config"]["cfg5"]["per_rank_stage_ms"][0]["semantic"] == 0 and d["config"]["cfg5"]["per_rank_stage_ms"][0]["fuse"] > 0
    assert d["config"]["host_cloud_frame_ms"] > 0 and d["config"]["cfg5"]["host_cloud_frame_ms"] > 0      # the same frames from a host cloud (bucketing + PCIe inclusive)
    sv = d["scaling_value"]                                                       # the strong-scaling figure, top level
    assert sv["speedup_vs_n1"] == d["config"]["cfg5"]["speedup_vs_n1"] and sv["rccl_ranks"] == 1 and sv["n_gpus"] == 1
    assert d["config"]["cfg4"]["rays"] in ("by row", "by ray over an all-reduced window")
    assert d["config"]["cfg4"]["stage_ms_rank0"]["rays"] > 0
