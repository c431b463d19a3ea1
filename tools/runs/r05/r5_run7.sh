# round 5, GPU call 7: quiet count in k_ray_apply, bucketed by-ray ownership
O=gpurun_out/r5g; mkdir -p $O
(time timeout 1500 python -m pytest tests/test_hip_bucketed.py tests/test_hip_terrain.py tests/test_hip_fullsize.py tests/test_hip_parity.py tests/test_hip_comm.py tests/test_hip_soak.py tests/test_hip_strips.py tests/test_hip_fuzz.py -m gpu -q -x) > $O/pytest.log 2>&1
tail -4 $O/pytest.log | grep -E "passed|failed|error"; grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | head -20
timeout 300 python bench.py --no-cpu-baseline --no-large > $O/default.json 2>> $O/err.log
python - <<PY
import json, os
d = json.load(open("$O/default.json")); c = d["config"]
print("default %.4f ms/step" % d["ms_per_step"], {k: round(v * 1e3, 1) for k, v in d["roofline"]["stage_ms"].items() if v > 0})
print("  cfg3:", c["cfg3"]["ms_per_step"], {k: round(v * 1e3, 1) for k, v in c["cfg3"]["stage_ms"].items() if v > 0}, "cold", c["cfg3"]["cold_start_ms"]["max_over_median"])
print("  terrain:", c["cfg3"]["terrain"]["ms_per_step"], {k: round(v * 1e3, 1) for k, v in c["cfg3"]["terrain"]["stage_ms"].items()}, "cold", c["cfg3"]["terrain"]["cold_start_ms"])
PY
tail -n 3 $O/err.log
