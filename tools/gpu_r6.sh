#!/bin/bash
# round 6 GPU helper.  usage: tools/gpu_r6.sh <tag> "<pytest args or empty>" [bench specs...]
# a bench spec is "label|ENV=.. ENV2=..|bench args" (env part may be empty); every spec runs REPS times (default 2)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/$1; mkdir -p $O; shift
PT="$1"; shift
if [ -n "$PT" ]; then
  (time timeout ${TEST_TIMEOUT:-1500} python -m pytest $PT -m gpu -q -x) > $O/pytest.log 2>&1
  grep -vE "^RCCL|^HIP ver|^ROCm|^Hostname|^Librccl|WARNING\] Layer|not found, adding" $O/pytest.log | tail -${TAIL:-25}
fi
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']
    print('%-28s' % '$1', '%.4f ms' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in r['stage_ms'].items() if v > 0.006})
except Exception as e: print('$1', 'FAILED', e)"; }
for spec in "$@"; do
  label=${spec%%|*}; rest=${spec#*|}; envs=${rest%%|*}; args=${rest#*|}
  for rep in $(seq 1 ${REPS:-2}); do
    env $envs timeout 900 python bench.py $args --no-cpu-baseline 2>>$O/err.log | tee $O/${label// /_}_$rep.json | line "$label"
  done
done
[ -f $O/err.log ] && tail -5 $O/err.log
exit 0
