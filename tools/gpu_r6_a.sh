#!/bin/bash
# round 6, milestone A on the GPU box: parity of the frames that carry their semantic channels, then cfg5 / cfg2 A/B against the
# round-5 library (tools/ab/r05.so) on the same box.  usage: tools/gpu_r6_a.sh [tag]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/${1:-r6a}; mkdir -p $O
(time timeout 1500 python -m pytest tests/test_hip_frame_semantics.py tests/test_hip_semantic.py tests/test_hip_parity.py tests/test_hip_terrain.py tests/test_hip_strips.py -m gpu -q -x) > $O/pytest.log 2>&1
grep -vE "^RCCL|^HIP ver|^ROCm|^Hostname|^Librccl" $O/pytest.log | tail -25
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1', '%.4f ms' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in r['stage_ms'].items() if v > 0})"; }
for rep in 1 2; do
  timeout 600 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline 2>>$O/err.log | tee $O/cfg5_new_$rep.json | line "cfg5 in-frame      "
  EMAP_HIP_LIB=$R/tools/ab/r05.so timeout 600 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline 2>>$O/err.log | tee $O/cfg5_r05_$rep.json | line "cfg5 r05 library   "
done
EMAP_BENCH_SEM_SEPARATE=1 timeout 600 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline 2>>$O/err.log | tee $O/cfg5_sep.json | line "cfg5 separate call "
EMAP_SEM_CARRY=0 timeout 600 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline 2>>$O/err.log | tee $O/cfg5_nocarry.json | line "cfg5 no carry      "
for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-cfg3 2>>$O/err.log | tee $O/cfg2_new_$rep.json | line "cfg2 new           "
  EMAP_HIP_LIB=$R/tools/ab/r05.so python bench.py --no-cpu-baseline --no-cfg3 2>>$O/err.log | tee $O/cfg2_r05_$rep.json | line "cfg2 r05 library   "
done
tail -5 $O/err.log
