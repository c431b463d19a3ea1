# GPU box: kernel trace of tools/exp_shift_trace.py -> gpurun_out/shift_trace/
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/shift_trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in 1 0; do EXP_MOVE=$m rocprofv3 --kernel-trace --stats -d $O/trace_move$m -- python $R/tools/exp_shift_trace.py > $O/run$m.log 2>&1; tail -1 $O/run$m.log; done
