O=gpurun_out/r4h; mkdir -p $O
for rep in 1 2; do for f in tools/ab/*.so; do
  EMAP_HIP_LIB=$PWD/$f timeout 300 python bench.py --workload cfg3 --steps 20 --no-cpu-baseline 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('$f cfg3', '%.4f ms' % d['ms_per_step'], 'rays', round(r['stage_ms']['rays']*1e3,1), 'cold', c['cold_start_ms']['max'], c['cold_start_ms']['median_last5'], c['cold_start_ms']['max_over_median'], c['cold_start_ms']['rays_ms_per_frame'][:4])"
done; done
for f in tools/ab/*.so; do
  EMAP_HIP_LIB=$PWD/$f timeout 300 python bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('$f cfg4', '%.4f ms' % d['ms_per_step'], 'rays', round(r['stage_ms']['rays']*1e3,1), 'cold', c['cold_start_ms']['max'], c['cold_start_ms']['median_last5'], c['cold_start_ms']['max_over_median'])"
done
timeout 300 python bench.py --workload cfg3 --steps 20 --no-cpu-baseline --sort-clouds angle 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('angle-sorted cfg3', '%.4f ms' % d['ms_per_step'], 'rays', round(r['stage_ms']['rays']*1e3,1), r['ray_visits_per_s'])"
