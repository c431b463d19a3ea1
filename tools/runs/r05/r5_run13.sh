# round 5, GPU call 13: histogram pass with four loads in flight (in-tree) vs one (tools/ab/hist_u1.so)
O=gpurun_out/r5m; mkdir -p $O
(time timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_randomized.py tests/test_hip_fuzz.py tests/test_hip_large_maps.py -k "not 8192 and not config5" -m gpu -q -x) > $O/pytest.log 2>&1
tail -4 $O/pytest.log | grep -E "passed|failed|error"; grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | head -20
for rep in 1 2 3; do
  for v in intree hist_u1; do
    L=""; [ $v != intree ] && L=$PWD/tools/ab/$v.so
    EMAP_HIP_LIB=$L python bench.py --no-cpu-baseline --no-cfg3 --no-large 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('cfg2 $v', '%.4f ms' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in r['stage_ms'].items() if k in ('hist','scan','scatter','gate','fuse','post')})"
  done
done
for v in intree hist_u1; do
  L=""; [ $v != intree ] && L=$PWD/tools/ab/$v.so
  EMAP_HIP_LIB=$L timeout 300 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('cfg5 $v', '%.4f ms' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in r['stage_ms'].items() if k in ('hist','scan','scatter','gate','fuse','post')})"
done
