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
  rocprofv3 --kernel-trace --stats -d $O/${wl}_trace -- python $R/bench.py --workload $wl --steps $steps --warmup 3 --no-cpu-baseline > $O/${wl}_trace.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    n=$(echo $c | cut -d" " -f1)
    rocprofv3 --kernel-trace --pmc $c -d $O/${wl}_pmc_$n -- python $R/bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline > $O/${wl}_pmc_$n.log 2>&1
  done
done
ls -R $O | head -40
# bench lines of the round (one JSON line each): default (cfg2 + config.cfg3), cfg3, shifted map, robot scale, cfg5 on one GPU
python $R/bench.py > $O/bench_cfg2.json 2>> $O/bench_err.log
python $R/bench.py --workload cfg3 --steps 20 > $O/bench_cfg3.json 2>> $O/bench_err.log
python $R/bench.py --pre-shift 37 21 --no-cpu-baseline > $O/bench_cfg2_shifted.json 2>> $O/bench_err.log
python $R/bench.py --cell-n 202 --points 50000 --no-cpu-baseline --no-cfg3 > $O/bench_cfg1.json 2>> $O/bench_err.log
timeout 600 python $R/bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_cfg4.json 2>> $O/bench_err.log
timeout 600 python $R/bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_cfg5.json 2>> $O/bench_err.log
tail -3 $O/bench_err.log
