# round-4 GPU call C: gpu suite, cfg5 with de-interleaved / interleaved clouds, strip emulation cfg5
O=gpurun_out/r4c; mkdir -p $O
(time timeout 1200 python -m pytest tests -m gpu -q -x) > $O/pytest.log 2>&1
tail -5 $O/pytest.log | grep -E "passed|failed|error"
grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | head -30
timeout 600 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline 2> $O/cfg5_split.err | head -1 > $O/cfg5_split.json
timeout 600 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline --interleaved-cloud 2> $O/cfg5_rows.err | head -1 > $O/cfg5_rows.json
python - <<PY
import json
for w in ("split", "rows"):
    d = json.load(open("$O/cfg5_%s.json" % w)); r = d["roofline"]
    print(w, "%.4f ms/step" % d["ms_per_step"], "%.0f Mpts/s" % d["value"], {k: round(v * 1e3, 1) for k, v in r["stage_ms"].items()}, "frac", r["frac"], r["kernel"])
PY
timeout 900 python tools/strip_emulation.py --workload cfg5 --steps 10 --gs 1 2 4 8 2> $O/strips_cfg5.err | head -1 > $O/strips_cfg5.json
python - <<PY
import json
d = json.load(open("$O/strips_cfg5.json"))
print("single", d["single"])
for g, sp in d["splits"].items():
    print(" G", g, {k: v for k, v in sp.items() if k not in ("rows", "stage_ms_net_rank0")})
print(d["wire"]["projected_speedup"])
PY
