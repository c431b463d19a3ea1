# round 5, GPU call 5: suite after the unchanged-cell store skip + DMA mask search; cfg5 / cfg4 lines; emulations (cfg4 by ray, terrain on 8 strips)
O=gpurun_out/r5e; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_hip_large_maps.py --deselect tests/test_hip_large_strips.py) > $O/pytest.log 2>&1
tail -4 $O/pytest.log | grep -E "passed|failed|error"; grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | head -20
for wl in cfg5 cfg4; do
  timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline > $O/$wl.json 2>> $O/err.log
done
timeout 300 python bench.py --no-cpu-baseline --no-large > $O/default.json 2>> $O/err.log
timeout 900 python tools/strip_emulation.py --workload cfg4 --rays --gs 8 --steps 10 2> $O/strips_cfg4.err | head -1 > $O/strips_cfg4_rays_by_ray.json
timeout 900 python tools/strip_emulation.py --workload cfg2 --rays --scene terrain --gs 8 --steps 10 2> $O/strips_terrain.err | head -1 > $O/strips_cfg3_terrain.json
python - <<PY
import json, os
for f in ("cfg5", "cfg4", "default"):
    p = "$O/%s.json" % f
    if not os.path.exists(p) or not os.path.getsize(p): print(f, "missing"); continue
    d = json.load(open(p)); r = d["roofline"]
    print(f, "%.4f ms/step" % d["ms_per_step"], {k: round(v * 1e3, 1) for k, v in r["stage_ms"].items() if v > 0}, "frac", r["frac"], r["kernel"])
    c = d["config"]
    if "cfg3" in c:
        print("  cfg3:", c["cfg3"]["ms_per_step"], {k: round(v * 1e3, 1) for k, v in c["cfg3"]["stage_ms"].items() if v > 0})
        if "terrain" in c["cfg3"]: print("  terrain:", c["cfg3"]["terrain"]["ms_per_step"], {k: round(v * 1e3, 1) for k, v in c["cfg3"]["terrain"]["stage_ms"].items()})
for f in ("strips_cfg4_rays_by_ray", "strips_cfg3_terrain"):
    p = "$O/%s.json" % f
    try: d = json.load(open(p))
    except Exception as e: print(f, "unreadable", e); continue
    print(f, "single", d["single"]["frame_ms"], d.get("ray_mode"), d.get("cloud"))
    for g, sp in d["splits"].items():
        print("  G", g, {k: sp.get(k) for k in ("solo_frame_ms_per_rank", "speedup_solo_no_wire", "cloud_share_per_rank", "lockstep_frame_ms_all_ranks_one_gpu", "lockstep_ms_per_rank_average", "speedup_bound_from_lockstep")})
        if "stage_ms_net_per_rank" in sp: print("   fuse per rank", [s_["fuse"] for s_ in sp["stage_ms_net_per_rank"]], "gate", [s_["gate"] for s_ in sp["stage_ms_net_per_rank"]], "rays", [s_["rays"] for s_ in sp["stage_ms_net_per_rank"]])
PY
tail -3 $O/err.log $O/strips_cfg4.err $O/strips_terrain.err
