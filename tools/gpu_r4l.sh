O=gpurun_out/r4l; mkdir -p $O
(time timeout 1200 python -m pytest tests -m gpu -q -x) > $O/pytest.log 2>&1
tail -5 $O/pytest.log | grep -E "passed|failed|error"
grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | head -30
for rep in 1 2; do for V in "--serial-semantic" ""; do
  timeout 300 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline $V 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
st=r['stage_ms']
print('cfg5 [$V]', '%.4f ms' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in st.items() if v > 0.01})"
done; done
for V in "--serial-semantic" ""; do
timeout 900 python tools/strip_emulation.py --workload cfg5 --steps 10 --gs 1 8 $V 2> $O/strips_cfg5.err | head -1 > $O/strips_cfg5$V.json
python - <<PY
import json
d = json.load(open("$O/strips_cfg5$V.json"))
print("[$V] single", d["single"]["frame_ms"])
for g, sp in d["splits"].items():
    print(" G", g, {k: v for k, v in sp.items() if k not in ("rows", "stage_ms_net_rank0", "stage_ms_net_slowest_rank")})
print(d["wire"]["projected_speedup"])
PY
done
