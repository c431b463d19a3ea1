#!/bin/bash
# final run of the round on the GPU box: priority parity tests, the round's profiles (tools/profile_round.sh without its bench lines),
# two bench lines, then the remaining tests as far as the time allows
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=r04
O=$R/gpurun_out/prof_$TAG
mkdir -p $O $R/gpurun_out/final
cd $R
PRIO="tests/test_hip_terrain.py tests/test_hip_parity.py tests/test_hip_fuzz.py tests/test_hip_randomized.py tests/test_hip_semantic.py tests/test_hip_strips.py tests/test_hip_comm.py tests/test_hip_large_maps.py tests/test_hip_shift.py tests/test_hip_fullsize.py"
timeout 400 python -m pytest $PRIO -m gpu -q -x 2>&1 | grep -vE "^RCCL|^HIP ver|^ROCm|^Hostname|^Librccl" | tail -5 > gpurun_out/final/tests_priority.txt
python -c "import bench; print(bench.source_stamp())" > $O/source_stamp.txt
cd /tmp && export TMPDIR=/tmp
for wl in cfg2 cfg3; do
  steps=20; [ $wl = cfg3 ] && steps=8
  rm -rf $O/${wl}_trace $O/${wl}_pmc_*
  rocprofv3 --kernel-trace --stats -d $O/${wl}_trace -- python $R/bench.py --workload $wl --steps $steps --warmup 3 --no-cpu-baseline --no-large --no-terrain > $O/${wl}_trace.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    n=$(echo $c | cut -d" " -f1)
    rocprofv3 --kernel-trace --pmc $c -d $O/${wl}_pmc_$n -- python $R/bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-large --no-terrain > $O/${wl}_pmc_$n.log 2>&1
  done
done
cd $R
python bench.py > $O/bench_cfg2.json 2>> $O/bench_err.log
python bench.py --workload cfg3 --steps 20 > $O/bench_cfg3.json 2>> $O/bench_err.log
IGN=""; for f in $PRIO; do IGN="$IGN --ignore=$f"; done
timeout ${REST_TIMEOUT:-400} python -m pytest tests -m gpu -q -x $IGN 2>&1 | grep -vE "^RCCL|^HIP ver|^ROCm|^Hostname|^Librccl" | tail -5 > gpurun_out/final/tests_rest.txt
cat gpurun_out/final/tests_priority.txt gpurun_out/final/tests_rest.txt; head -c 400 $O/bench_cfg2.json
