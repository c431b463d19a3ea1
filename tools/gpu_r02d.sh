O=gpurun_out/r02d; mkdir -p $O
(time python -m pytest -m gpu -x -q tests/test_hip_parity.py tests/test_hip_fullsize.py tests/test_hip_strips.py tests/test_hip_warm_fixtures.py tests/test_hip_randomized.py) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log
for lbm in 1; do
  EMAP_RAY_LDS_BITMAP=$lbm python bench.py --workload cfg3 --steps 20 --no-cpu-baseline > $O/cfg3_lbm$lbm.json 2>> $O/err.log
  python - <<PY
import json
d=json.load(open("$O/cfg3_lbm$lbm.json")); r=d["roofline"]["stage_ms"]
print("lbm $lbm: %.4f ms/step" % d["ms_per_step"], r, d["roofline"]["ray_visits_per_frame"])
PY
done
EMAP_RAY_LDS_BITMAP=1 python bench.py --workload cfg3 --steps 20 --no-cpu-baseline --sort-clouds angle > $O/cfg3_angle.json 2>> $O/err.log
python -c "
import json; d=json.load(open('$O/cfg3_angle.json')); print('angle-sorted', d['ms_per_step'], d['roofline']['stage_ms']['rays'])"
python -c "
tail -3 $O/err.log
