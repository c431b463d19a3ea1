# round 5, GPU call 11: coarse visit thresholds in k_rays: parity (rays tests) + A/B against the committed library (tools/ab/head.so) and with the filter off
O=gpurun_out/r5k; mkdir -p $O
(time timeout 1500 python -m pytest tests/test_hip_fullsize.py tests/test_hip_parity.py tests/test_hip_terrain.py tests/test_hip_shift.py tests/test_hip_strips.py tests/test_hip_comm.py tests/test_hip_fuzz.py tests/test_hip_randomized.py tests/test_hip_normals_exact.py tests/test_hip_soak.py -m gpu -q -x) > $O/pytest.log 2>&1
tail -4 $O/pytest.log | grep -E "passed|failed|error"; grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | head -20
for rep in 1 2; do
  for v in intree head off; do
    L=""; E="X=0"; [ $v = head ] && L=$PWD/tools/ab/head.so; [ $v = off ] && E="EMAP_RAY_COARSE=0"
    env $E EMAP_HIP_LIB=$L python bench.py --workload cfg3 --steps 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('cfg3 $v', '%.4f ms' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in r['stage_ms'].items() if k in ('fuse','rays','average','post')}, 'cold', d['config']['cold_start_ms']['max_over_median'], d['config']['cold_start_ms']['max'])"
  done
done
for v in intree head; do
  L=""; [ $v = head ] && L=$PWD/tools/ab/head.so
  EMAP_HIP_LIB=$L timeout 300 python bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('cfg4 $v', '%.4f ms' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in r['stage_ms'].items() if k in ('fuse','rays','average','post')})"
  EMAP_HIP_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-large 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('default $v', '%.4f ms' % d['ms_per_step'], 'cfg3', c['cfg3']['ms_per_step'], 'terrain', c['cfg3']['terrain']['ms_per_step'], {k: round(v*1e3,1) for k,v in c['cfg3']['terrain']['stage_ms'].items() if k in ('fuse','rays','post')}, 'cfg1', c['cfg1']['rays_overlap']['ms_per_step'])"
done
