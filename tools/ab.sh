# GPU box: A/B of library builds under tools/ab/ on the cfg2 (and optionally cfg3) bench.  usage: tools/ab.sh [cfg3]
for f in tools/ab/*.so; do
  for rep in 1 2; do
    EMAP_HIP_LIB=$PWD/$f python bench.py --no-cpu-baseline --no-cfg3 ${1:+--workload $1 --steps 20} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$f', '%.4f ms' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in r['stage_ms'].items() if k in ('gate','fuse','rays','post','hist','scan','scatter','overlap')})"
  done
done
