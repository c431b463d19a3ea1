# round 5, GPU call 1: new parity tests + fast suite, then A/B of k_post_pipe on cfg4 / cfg5 and the default line
O=gpurun_out/r5a; mkdir -p $O
(time timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_hip_large_maps.py --deselect tests/test_hip_large_strips.py) > $O/pytest.log 2>&1
tail -4 $O/pytest.log | grep -E "passed|failed|error"; grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | head -20
for pp in 0 1; do
  for wl in cfg5 cfg4; do
    EMAP_POST_PIPE=$pp timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline > $O/${wl}_pipe$pp.json 2>> $O/err.log
  done
done
timeout 600 python bench.py --no-cpu-baseline > $O/default.json 2>> $O/err.log
python - <<PY
import json, os
for f in ("cfg5_pipe0", "cfg5_pipe1", "cfg4_pipe0", "cfg4_pipe1", "default"):
    p = "$O/%s.json" % f
    if not os.path.exists(p) or not os.path.getsize(p): print(f, "missing"); continue
    d = json.load(open(p)); r = d["roofline"]
    print(f, "%.4f ms/step" % d["ms_per_step"], {k: round(v * 1e3, 1) for k, v in r["stage_ms"].items() if v > 0}, "frac", r["frac"], r["kernel"])
    c = d["config"]
    if "cfg3" in c:
        print("  cfg3:", c["cfg3"]["ms_per_step"], {k: round(v * 1e3, 1) for k, v in c["cfg3"]["stage_ms"].items() if v > 0}, "cold", c["cfg3"]["cold_start_ms"])
        if "terrain" in c["cfg3"]: print("  terrain:", c["cfg3"]["terrain"]["ms_per_step"], {k: round(v * 1e3, 1) for k, v in c["cfg3"]["terrain"]["stage_ms"].items()}, "cold", c["cfg3"]["terrain"]["cold_start_ms"])
    for k in ("cfg1", "cfg4", "cfg5"):
        if k in c: print("  %s:" % k, json.dumps(c[k])[:600])
PY
tail -5 $O/err.log
