# GPU box: A/B of environment knobs on one bench workload.  usage: tools/ab_env.sh "<bench args>" "VAR=a VAR2=b" "VAR=c" ...
ARGS="$1"; shift
for cfg in "$@"; do
  for rep in 1 2; do
    env $cfg python bench.py --no-cpu-baseline --no-cfg3 $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-40s' % '$cfg', '%.4f ms' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in r['stage_ms'].items() if k in ('gate','fuse','rays','post','hist','scan','scatter')})"
  done
done
