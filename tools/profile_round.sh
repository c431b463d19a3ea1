#!/bin/bash
# Run ON THE GPU BOX (gpurun): kernel-trace stats + separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ) of bench.py for both
# workloads; results land in gpurun_out/prof_<tag>/ and are converted locally with tools/rocprof_summary.py / pmc_to_json.py.
# usage: tools/profile_round.sh <tag>
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
(cd $R && python -c "import bench; print(bench.source_stamp())") > $O/source_stamp.txt      # which kernel sources these databases belong to
cd /tmp && export TMPDIR=/tmp
for wl in cfg2 cfg3; do
  steps=20; [ $wl = cfg3 ] && steps=8
  rocprofv3 --kernel-trace --stats -d $O/${wl}_trace -- python $R/bench.py --workload $wl --steps $steps --warmup 3 --no-cpu-baseline --no-large > $O/${wl}_trace.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    n=$(echo $c | cut -d" " -f1)
    rocprofv3 --kernel-trace --pmc $c -d $O/${wl}_pmc_$n -- python $R/bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-large > $O/${wl}_pmc_$n.log 2>&1
  done
done
ls -R $O | head -40
bash $R/tools/bench_lines.sh $TAG
