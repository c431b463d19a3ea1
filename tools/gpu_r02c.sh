O=gpurun_out/r02c; mkdir -p $O
(time python -m pytest tests -m gpu -x -q) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log
for idx in 2 1; do
  EMAP_RAY_IDX=$idx python bench.py --workload cfg3 --steps 20 --no-cpu-baseline > $O/cfg3_idx$idx.json 2>> $O/err.log
  python - <<PY
import json
d=json.load(open("$O/cfg3_idx$idx.json")); r=d["roofline"]["stage_ms"]
print("idx $idx: %.4f ms/step" % d["ms_per_step"], r, d["roofline"]["ray_visits_per_frame"])
PY
done
python bench.py --no-cpu-baseline > $O/cfg2.json 2>> $O/err.log; python -c "
import json; d=json.load(open('$O/cfg2.json')); print('cfg2', d['ms_per_step'], d['roofline']['stage_ms']); print(d['config']['cfg3'])"
tail -3 $O/err.log
