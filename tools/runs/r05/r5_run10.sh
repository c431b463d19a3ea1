# round 5, GPU call 10: environment A/B on the final kernels (stencil tile height, sort chunk)
for cfg in "X=0" "EMAP_POST_R=16" "EMAP_BIN_CHUNK=2048" "EMAP_BIN_CHUNK=8192"; do
  for rep in 1 2; do
    env $cfg python bench.py --no-cpu-baseline --no-cfg3 --no-large 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-24s' % '$cfg', '%.4f ms' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in r['stage_ms'].items() if k in ('hist','scan','scatter','gate','fuse','post')})"
  done
done
