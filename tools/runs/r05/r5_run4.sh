# round 5, GPU call 4: suite after the mask search / 32-byte window records, default bench line (terrain post stage), cfg5 line
O=gpurun_out/r5d; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_hip_large_maps.py --deselect tests/test_hip_large_strips.py) > $O/pytest.log 2>&1
tail -4 $O/pytest.log | grep -E "passed|failed|error"; grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | head -20
timeout 600 python bench.py --no-cpu-baseline > $O/default.json 2>> $O/err.log
python - <<PY
import json, os
for f in ("default",):
    p = "$O/%s.json" % f
    if not os.path.exists(p) or not os.path.getsize(p): print(f, "missing"); continue
    d = json.load(open(p)); r = d["roofline"]
    print(f, "%.4f ms/step" % d["ms_per_step"], {k: round(v * 1e3, 1) for k, v in r["stage_ms"].items() if v > 0}, "frac", r["frac"], r["kernel"])
    c = d["config"]
    if "cfg3" in c:
        print("  cfg3:", c["cfg3"]["ms_per_step"], {k: round(v * 1e3, 1) for k, v in c["cfg3"]["stage_ms"].items() if v > 0}, "cold", c["cfg3"]["cold_start_ms"]["max_over_median"])
        if "terrain" in c["cfg3"]: print("  terrain:", c["cfg3"]["terrain"]["ms_per_step"], {k: round(v * 1e3, 1) for k, v in c["cfg3"]["terrain"]["stage_ms"].items()}, "cold", c["cfg3"]["terrain"]["cold_start_ms"]["max_over_median"])
    for k in ("cfg4", "cfg5"):
        if k in c: print("  %s:" % k, c[k]["ms_per_step"], c[k]["stage_ms"])
PY
tail -5 $O/err.log
