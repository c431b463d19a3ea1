# round 5, GPU call 17: does the global-bitmap ray kernel (two workgroups per CU) start a cleared map faster?  (EMAP_RAY_LMAP=0 forces it for every frame)
summ='
import json,sys
d=json.loads(sys.stdin.read()); c=d["config"]["cfg3"]
print(sys.argv[1], json.dumps({"ms":c["ms_per_step"],"rays":round(c["stage_ms"]["rays"]*1e3,1),"cold":c["cold_start_ms"]}))'
for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-large --no-terrain 2>/dev/null | python -c "$summ" default
  EMAP_RAY_LMAP=0 timeout 300 python bench.py --no-cpu-baseline --no-large --no-terrain 2>/dev/null | python -c "$summ" lmap0
done
