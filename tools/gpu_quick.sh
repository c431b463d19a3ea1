# GPU box: the full gpu suite (minus the slow large-map tests unless LARGE=1), then bench lines.  usage: tools/gpu_quick.sh <tag> [pytest -k expr]
O=gpurun_out/${1:-quick}; mkdir -p $O
K=${2:-}
DESEL="--deselect tests/test_hip_large_maps.py"; [ "${LARGE:-0}" = 1 ] && DESEL=""
(time python -m pytest tests -m gpu -q -x $DESEL ${K:+-k "$K"}) > $O/pytest.log 2>&1
tail -5 $O/pytest.log | grep -E "passed|failed|error"
grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | head -30
python bench.py --no-cpu-baseline > $O/cfg2.json 2>> $O/err.log
[ "${CFG5:-0}" = 1 ] && timeout 600 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $O/cfg5.json 2>> $O/err.log
python - <<PY
import json, os
for w in ("cfg2", "cfg5"):
    f = "$O/%s.json" % w
    if not os.path.exists(f) or not os.path.getsize(f): continue
    d = json.load(open(f)); r = d["roofline"]
    print(w, "%.4f ms/step" % d["ms_per_step"], "%.0f Mpts/s" % d["value"], {k: round(v * 1e3, 1) for k, v in r["stage_ms"].items()}, "frac", r["frac"], r["kernel"])
    if "cfg3" in d["config"]: print("  config.cfg3:", d["config"]["cfg3"]["ms_per_step"], d["config"]["cfg3"]["value"], {k: round(v * 1e3, 1) for k, v in d["config"]["cfg3"]["stage_ms"].items()})
PY
tail -3 $O/err.log
