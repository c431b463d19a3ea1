#!/bin/bash
# closing run of a round on the GPU box: the round's profiles (tools/profile_round.sh without its bench lines; the terrain
# sub-measurement is skipped under the profiler), then the parity tests.  Afterwards, locally: tools/profiles_from_gpurun.sh <tag>,
# then tools/bench_lines.sh <tag> on the GPU box (the lines then quote the stamped kernel statistics) and the import once more.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O $R/gpurun_out/final
cd $R
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
timeout ${TEST_TIMEOUT:-900} python -m pytest ${TESTS:-tests} -m gpu -q -x 2>&1 | grep -vE "^RCCL|^HIP ver|^ROCm|^Hostname|^Librccl" | tail -6 > gpurun_out/final/tests.txt
cat gpurun_out/final/tests.txt
