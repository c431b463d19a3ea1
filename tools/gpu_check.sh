# GPU box: full gpu test suite, then the default bench line (cfg2 + config.cfg3) and the cfg3 line
O=gpurun_out/${1:-check}; mkdir -p $O
(time python -m pytest tests -m gpu -x -q) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log
python bench.py --no-cpu-baseline > $O/cfg2.json 2>> $O/err.log
python bench.py --workload cfg3 --steps 20 --no-cpu-baseline > $O/cfg3.json 2>> $O/err.log
python - <<PY
import json
for w in ("cfg2", "cfg3"):
    d = json.load(open("$O/%s.json" % w)); r = d["roofline"]
    print(w, "%.4f ms/step" % d["ms_per_step"], "%.0f Mpts/s" % d["value"], {k: round(v * 1e3, 1) for k, v in r["stage_ms"].items()}, "frac", r["frac"], r["kernel"])
    if "cfg3" in d["config"]: print("  config.cfg3:", d["config"]["cfg3"]["ms_per_step"], d["config"]["cfg3"]["value"])
PY
tail -3 $O/err.log
