# round 5, GPU call 6: ray-kernel preference from the quiet share, bucketed clouds on the plain sort kernels
O=gpurun_out/r5f; mkdir -p $O
(time timeout 1500 python -m pytest tests/test_hip_bucketed.py tests/test_hip_terrain.py tests/test_hip_fullsize.py tests/test_hip_parity.py tests/test_hip_comm.py tests/test_hip_soak.py -m gpu -q -x) > $O/pytest.log 2>&1
tail -4 $O/pytest.log | grep -E "passed|failed|error"; grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | head -20
timeout 300 python bench.py --no-cpu-baseline --no-large > $O/default.json 2>> $O/err.log
EMAP_RAY_LMAP=1 timeout 300 python bench.py --no-cpu-baseline --no-large > $O/default_lmap1.json 2>> $O/err.log
timeout 900 python tools/strip_emulation.py --workload cfg5 --gs 8 --steps 10 --no-lockstep 2> $O/strips_cfg5.err | head -1 > $O/strips_cfg5.json
python - <<PY
import json, os
for f in ("default", "default_lmap1"):
    p = "$O/%s.json" % f
    if not os.path.exists(p) or not os.path.getsize(p): print(f, "missing"); continue
    d = json.load(open(p)); r = d["roofline"]
    print(f, "%.4f ms/step" % d["ms_per_step"])
    c = d["config"]
    if "cfg3" in c:
        print("  cfg3:", c["cfg3"]["ms_per_step"], {k: round(v * 1e3, 1) for k, v in c["cfg3"]["stage_ms"].items() if v > 0}, "cold", c["cfg3"]["cold_start_ms"])
        if "terrain" in c["cfg3"]: print("  terrain:", c["cfg3"]["terrain"]["ms_per_step"], {k: round(v * 1e3, 1) for k, v in c["cfg3"]["terrain"]["stage_ms"].items()}, "cold", c["cfg3"]["terrain"]["cold_start_ms"])
    if "cfg1" in c: print("  cfg1", json.dumps(c["cfg1"])[:400])
d = json.load(open("$O/strips_cfg5.json"))
print("single", d["single"]["frame_ms"])
for g, sp in d["splits"].items():
    print("G", g, sp.get("solo_frame_ms_per_rank"), sp.get("speedup_solo_no_wire"), "slowest", sp.get("stage_ms_net_slowest_rank"))
PY
tail -n 3 $O/err.log $O/strips_cfg5.err
