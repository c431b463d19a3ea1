# round 5, GPU call 2: suite after the general normal-row exchange; k_post_pipe variants on cfg5
O=gpurun_out/r5b; mkdir -p $O
(time timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_hip_large_maps.py --deselect tests/test_hip_large_strips.py) > $O/pytest.log 2>&1
tail -4 $O/pytest.log | grep -E "passed|failed|error"; grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | head -20
for v in pipe_occ6 pipe_occ6_nopf pipe_occ4_nopf; do
  EMAP_HIP_LIB=$PWD/tools/ab/$v.so EMAP_POST_PIPE=1 timeout 300 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $O/cfg5_$v.json 2>> $O/err.log
done
python - <<PY
import json, os
for f in ("cfg5_pipe_occ6", "cfg5_pipe_occ6_nopf", "cfg5_pipe_occ4_nopf"):
    p = "$O/%s.json" % f
    if not os.path.exists(p) or not os.path.getsize(p): print(f, "missing"); continue
    d = json.load(open(p)); r = d["roofline"]
    print(f, "%.4f ms/step" % d["ms_per_step"], {k: round(v * 1e3, 1) for k, v in r["stage_ms"].items() if v > 0})
PY
tail -5 $O/err.log
