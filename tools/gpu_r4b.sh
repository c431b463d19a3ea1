# round-4 GPU call B: full gpu suite (rays by ray), strip emulation of cfg4 with rays: by row vs by ray
O=gpurun_out/r4b; mkdir -p $O
(time timeout 1200 python -m pytest tests -m gpu -q -x) > $O/pytest.log 2>&1
tail -5 $O/pytest.log | grep -E "passed|failed|error"
grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | head -30
for M in by_row by_ray; do
  timeout 900 python tools/strip_emulation.py --workload cfg4 --rays --ray-mode $M --steps 6 --gs 1 4 8 > $O/strips_cfg4_rays_$M.json 2> $O/strips_cfg4_rays_$M.err
  tail -2 $O/strips_cfg4_rays_$M.err
  python - <<PY
import json
d = json.load(open("$O/strips_cfg4_rays_$M.json"))
print("$M single", d["single"]["frame_ms"], d["single"]["stage_ms_net"])
for g, sp in d["splits"].items():
    print(" G", g, {k: v for k, v in sp.items() if k not in ("rows", "stage_ms_net_rank0")})
print(d["wire"]["projected_speedup"])
PY
done
