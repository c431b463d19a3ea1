O=gpurun_out/r4j; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_comm.py tests/test_hip_strips.py tests/test_hip_large_strips.py tests/test_hip_multiproc.py -q -x 2>&1 | tail -3
timeout 900 python tools/strip_emulation.py --workload cfg5 --steps 10 --gs 1 8 2> $O/strips_cfg5.err | head -1 > $O/strips_cfg5.json
python - <<PY
import json
d = json.load(open("$O/strips_cfg5.json"))
print("single", d["single"])
for g, sp in d["splits"].items():
    print(" G", g, {k: v for k, v in sp.items() if k not in ("rows", "stage_ms_net_rank0")})
print(d["wire"]["projected_speedup"])
PY
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o strips -- python $GRAFT_REPO_ROOT/tools/strip_emulation.py --workload cfg5 --steps 5 --gs 8 --no-lockstep --skip-single > $GRAFT_REPO_ROOT/$O/prof.out 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT
ls -R $O/prof | head; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); echo $f; [ -n "$f" ] && head -25 $f | cut -c1-220
rm -f $(find $O/prof -name "*.db") 2>/dev/null
