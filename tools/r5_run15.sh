# round 5, GPU call 15: k_bin_sort (sort front-end in one launch): parity, then A/B against the three launches (EMAP_BIN_SORT=0)
O=gpurun_out/r5o; mkdir -p $O
(time timeout 1200 python -m pytest tests/test_hip_small_frame.py tests/test_hip_terrain.py tests/test_hip_parity.py tests/test_hip_soak.py tests/test_hip_shift.py tests/test_hip_semantic.py tests/test_hip_fuzz.py tests/test_hip_randomized.py tests/test_hip_warm_fixtures.py -m gpu -q -x) > $O/pytest.log 2>&1
tail -4 $O/pytest.log | grep -E "passed|failed|error"; grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | head -30
summ='
import json,sys
d=json.loads(sys.stdin.read()); c=d["config"]; r=d["roofline"]
out={"ms":d["ms_per_step"], "p50":c.get("latency_ms",{}).get("p50"), "dom":r["kernel"], "frac":r["frac"], "st":{k: round(v*1e3,1) for k,v in r["stage_ms"].items() if k in ("hist","scan","scatter","gate","fuse","post")}}
c3=c.get("cfg3")
if c3: out["cfg3"]={"ms":c3["ms_per_step"],"st":{k: round(v*1e3,1) for k,v in c3["stage_ms"].items() if v>0}}
t=c.get("cfg3",{}).get("terrain")
if t: out["terrain"]={"ms":t["ms_per_step"],"st":{k: round(v*1e3,1) for k,v in t["stage_ms"].items()}}
print(sys.argv[1], json.dumps(out))'
for rep in 1 2 3; do
  timeout 400 python bench.py --no-cpu-baseline --no-large 2>$O/err_new$rep.log | python -c "$summ" new
  EMAP_BIN_SORT=0 timeout 400 python bench.py --no-cpu-baseline --no-large 2>$O/err_old$rep.log | python -c "$summ" old
done
tail -3 $O/err_new1.log
