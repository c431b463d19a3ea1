#!/bin/bash
# Run ON THE GPU BOX: the bench lines of a round (one JSON line each) -> gpurun_out/prof_<tag>/bench_*.json.  Run it once more AFTER
# tools/profiles_from_gpurun.sh has written the stamped kernel statistics, so that roofline.frac of the committed lines is computed from
# profiles/<tag>_<workload>_kernel_stats.txt (kernel_us_rocprof) like the driver's own run.   usage: tools/bench_lines.sh <tag>
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd $R
# default (cfg2 + config.cfg3), cfg3, shifted map, robot scale, cfg4 and cfg5 on one GPU
python $R/bench.py > $O/bench_cfg2.json 2>> $O/bench_err.log
python $R/bench.py --workload cfg3 --steps 20 > $O/bench_cfg3.json 2>> $O/bench_err.log
python $R/bench.py --pre-shift 37 21 --no-cpu-baseline > $O/bench_cfg2_shifted.json 2>> $O/bench_err.log
python $R/bench.py --cell-n 202 --points 50000 --no-cpu-baseline --no-cfg3 > $O/bench_cfg1.json 2>> $O/bench_err.log
timeout 600 python $R/bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_cfg4.json 2>> $O/bench_err.log
timeout 600 python $R/bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_cfg5.json 2>> $O/bench_err.log
tail -3 $O/bench_err.log
