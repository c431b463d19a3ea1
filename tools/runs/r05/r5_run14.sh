# round 5, GPU call 14: k_small_frame (robot-scale frame in one launch) + standing pool of extra tile workgroups: parity, then A/B
O=gpurun_out/r5n; mkdir -p $O
(time timeout 900 python -m pytest tests/test_hip_small_frame.py tests/test_hip_terrain.py tests/test_hip_parity.py tests/test_hip_shift.py tests/test_hip_semantic.py tests/test_hip_fuzz.py tests/test_hip_randomized.py -m gpu -q -x) > $O/pytest.log 2>&1
tail -4 $O/pytest.log | grep -E "passed|failed|error"; grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | head -30
summ='
import json,sys
d=json.loads(sys.stdin.read()); c=d["config"]; r=d["roofline"]
out={"ms":d["ms_per_step"], "st":{k: round(v*1e3,1) for k,v in r["stage_ms"].items() if k in ("hist","scan","scatter","gate","fuse","post")}}
if "cfg1" in c: out["cfg1"]={k:(v["ms_per_step"], v["latency_ms"]["p50"], v.get("path")) for k,v in c["cfg1"].items() if isinstance(v,dict)}
t=c.get("cfg3",{}).get("terrain")
if t: out["terrain"]={"ms":t["ms_per_step"],"st":{k: round(v*1e3,1) for k,v in t["stage_ms"].items()},"chg":t.get("after_uniform_frames")}
print(sys.argv[1], json.dumps(out))'
for rep in 1 2; do
  timeout 400 python bench.py --no-cpu-baseline --no-large 2>$O/err_new$rep.log | python -c "$summ" new
  EMAP_SMALL_FRAME=0 EMAP_SPLIT_POOL=0 timeout 400 python bench.py --no-cpu-baseline --no-large 2>$O/err_old$rep.log | python -c "$summ" old
done
for v in 16 0; do
  EMAP_SPLIT_POOL=$v timeout 300 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$summ" cfg5_pool$v
done
tail -3 $O/err_new1.log
