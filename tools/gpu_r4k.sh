O=gpurun_out/r4k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_strips -o strips -- python $GRAFT_REPO_ROOT/tools/strip_emulation.py --workload cfg5 --steps 5 --gs 8 --no-lockstep --skip-single > $GRAFT_REPO_ROOT/$O/prof.out 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_strips -name "*.db" | head -1); echo $f
python tools/rocprof_summary.py $f > $O/strips_cfg5_solo_kernel_stats.txt
head -30 $O/strips_cfg5_solo_kernel_stats.txt | cut -c1-150
