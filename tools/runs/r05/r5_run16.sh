# round 5, GPU call 16: k_small_frame with the counter barrier (one non-returning atomic + poll) and the gate evaluated by every workgroup
O=gpurun_out/r5p; mkdir -p $O
(time timeout 900 python -m pytest tests/test_hip_small_frame.py tests/test_hip_parity.py tests/test_hip_shift.py -m gpu -q -x) > $O/pytest.log 2>&1
tail -4 $O/pytest.log | grep -E "passed|failed|error"; grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | head -30
summ='
import json,sys
d=json.loads(sys.stdin.read()); c=d["config"]; r=d["roofline"]
out={"ms":d["ms_per_step"], "st":{k: round(v*1e3,1) for k,v in r["stage_ms"].items() if k in ("hist","scan","scatter","gate","fuse","post")}}
if "cfg1" in c: out["cfg1"]={k:(v["ms_per_step"], v["latency_ms"]["p50"], v.get("path")) for k,v in c["cfg1"].items() if isinstance(v,dict)}
print(sys.argv[1], json.dumps(out))'
for rep in 1 2; do
  timeout 400 python bench.py --no-cpu-baseline --no-large --no-terrain 2>$O/err_new$rep.log | python -c "$summ" new
  EMAP_SMALL_FRAME=0 timeout 400 python bench.py --no-cpu-baseline --no-large --no-terrain 2>$O/err_old$rep.log | python -c "$summ" old
done
tail -3 $O/err_new1.log
