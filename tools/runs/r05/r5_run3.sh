# round 5, GPU call 3: bucketed clouds (parity + bench path), strip emulation of cfg5 at G = 8
O=gpurun_out/r5c; mkdir -p $O
(time timeout 1200 python -m pytest tests/test_hip_bucketed.py tests/test_bench_contract.py tests/test_hip_upload.py tests/test_hip_strips.py -m gpu -q -x) > $O/pytest.log 2>&1
tail -4 $O/pytest.log | grep -E "passed|failed|error"; grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | head -30
timeout 900 python tools/strip_emulation.py --workload cfg5 --gs 8 --steps 10 > $O/strips_cfg5.json 2> $O/strips_cfg5.err
python - <<PY
import json
d = json.load(open("$O/strips_cfg5.json"))
print("single", d["single"]["frame_ms"], d["single"]["stage_ms_net"])
for g, sp in d["splits"].items():
    print("G", g, {k: sp.get(k) for k in ("solo_frame_ms_per_rank", "solo_frame_ms_max", "speedup_solo_no_wire", "cloud_share_per_rank", "lockstep_frame_ms_all_ranks_one_gpu", "speedup_bound_from_lockstep")})
    print("  slowest", sp.get("stage_ms_net_slowest_rank"))
print(d["wire"]["projected_speedup"])
PY
tail -3 $O/strips_cfg5.err
