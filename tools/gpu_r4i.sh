O=gpurun_out/r4i; mkdir -p $O
(time timeout 1200 python -m pytest tests -m gpu -q -x) > $O/pytest.log 2>&1
tail -5 $O/pytest.log | grep -E "passed|failed|error"
grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | head -30
for rep in 1 2; do for f in tools/ab/*.so; do
  EMAP_HIP_LIB=$PWD/$f timeout 300 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
st=r['stage_ms']
print('$f cfg5', '%.4f ms' % d['ms_per_step'], 'semantic ~ %.3f' % (d['ms_per_step']-sum(st.values())+0.03), {k: round(v*1e3,1) for k,v in st.items() if v > 0.01})"
done; done
