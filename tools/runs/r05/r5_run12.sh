# round 5, GPU call 12: drift gate folded into k_count's last workgroup (small clouds): parity + robot-scale A/B
O=gpurun_out/r5l; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_hip_large_maps.py --deselect tests/test_hip_large_strips.py) > $O/pytest.log 2>&1
tail -4 $O/pytest.log | grep -E "passed|failed|error"; grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | head -20
for rep in 1 2 3; do
  for f in 1 0; do
    EMAP_GATE_FOLD=$f python bench.py --cell-n 202 --points 50000 --no-cpu-baseline --no-cfg3 --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('cfg1 fold=$f', '%.5f ms' % d['ms_per_step'], c['latency_ms'])"
    EMAP_GATE_FOLD=$f python bench.py --workload cfg3 --cell-n 202 --points 50000 --no-cpu-baseline --no-cfg3 --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('cfg1+rays fold=$f', '%.5f ms' % d['ms_per_step'], c['latency_ms'])"
  done
done
