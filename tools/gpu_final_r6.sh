#!/bin/bash
# Closing run of round 6 on the GPU box, ONE call: rocprofv3 passes (kernel trace + separate PMC passes) of cfg2 / cfg3 / cfg5, converted
# on the box into profiles/r06_* (so that the bench lines that follow quote the stamped kernel statistics, like the driver's own run),
# the bench lines, the strip emulations on the final kernels, then the whole GPU suite incl. the large maps.  Everything the repo keeps
# is copied under gpurun_out/prof_r06/ (merged back); locally: cp gpurun_out/prof_r06/profiles/* profiles/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=r06
O=$R/gpurun_out/prof_$TAG
mkdir -p $O $O/profiles
cd $R
python -c "import bench; print(bench.source_stamp())" > $O/source_stamp.txt
STAMP=$(cat $O/source_stamp.txt)
cd /tmp && export TMPDIR=/tmp
for wl in cfg2 cfg3 cfg5; do
  steps=20; extra="--no-large --no-terrain"; [ $wl = cfg3 ] && steps=8; [ $wl = cfg5 ] && steps=6
  rocprofv3 --kernel-trace --stats -d $O/${wl}_trace -- python $R/bench.py --workload $wl --steps $steps --warmup 3 --no-cpu-baseline $extra > $O/${wl}_trace.log 2>&1
  counters=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES")
  [ $wl = cfg5 ] && counters=("FETCH_SIZE" "WRITE_SIZE")
  for c in "${counters[@]}"; do
    n=$(echo $c | cut -d" " -f1)
    rocprofv3 --kernel-trace --pmc $c -d $O/${wl}_pmc_$n -- python $R/bench.py --workload $wl --steps 4 --warmup 2 --no-cpu-baseline $extra > $O/${wl}_pmc_$n.log 2>&1
  done
done
cd $R
for wl in cfg2 cfg3 cfg5; do
  (echo "# $TAG: rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl (tools/gpu_final_r6.sh)"; echo "# source_stamp: $STAMP"; python tools/rocprof_summary.py $(ls $O/${wl}_trace/*/*_results.db | head -1)) > profiles/${TAG}_${wl}_kernel_stats.txt
  python tools/pmc_to_json.py $(ls $O/${wl}_pmc_FETCH_SIZE/*/*_results.db | head -1) $(ls $O/${wl}_pmc_WRITE_SIZE/*/*_results.db | head -1) $STAMP > profiles/${TAG}_pmc_${wl}.json
  if [ $wl != cfg5 ]; then
  python - <<PY > profiles/${TAG}_${wl}_sq_counters.txt
import sqlite3,glob
f=glob.glob("$O/${wl}_pmc_SQ_WAVES/*/*_results.db")[0]
cur=sqlite3.connect(f).cursor()
rows=cur.execute("select kernel_name, counter_name, value from counters_collection order by kernel_name, counter_name, dispatch_id").fetchall()
by={}
for k,c,v in rows: by.setdefault((k,c),[]).append(v)
print("# $TAG $wl: rocprofv3 --pmc SQ_* (per dispatch: average, number of dispatches, MEDIAN -- the median is the steady state, the average includes the warm-up and cold-start frames)")
print("# source_stamp: $STAMP")
for (k,c),vs in sorted(by.items()):
    vs.sort(); n=len(vs); med=vs[n//2] if n%2 else 0.5*(vs[n//2-1]+vs[n//2])
    print("%-64s %-22s %16.1f n=%d med=%.1f"%(k[:64],c,sum(vs)/n,n,med))
PY
  fi
done
# bench lines (now quoting profiles/r06_*): default (cfg2 + config.cfg3 / terrain / cfg1 / cfg4 / cfg5), cfg3, shifted map, robot scale, cfg4, cfg5
python bench.py > profiles/${TAG}_bench_cfg2.json 2>> $O/bench_err.log
python bench.py --workload cfg3 --steps 20 > profiles/${TAG}_bench_cfg3.json 2>> $O/bench_err.log
python bench.py --pre-shift 37 21 --no-cpu-baseline --no-large > profiles/${TAG}_bench_cfg2_shifted.json 2>> $O/bench_err.log
python bench.py --cell-n 202 --points 50000 --no-cpu-baseline --no-cfg3 > profiles/${TAG}_bench_cfg1.json 2>> $O/bench_err.log
timeout 600 python bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu-baseline > profiles/${TAG}_bench_cfg4.json 2>> $O/bench_err.log
timeout 600 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline > profiles/${TAG}_bench_cfg5.json 2>> $O/bench_err.log
timeout 600 python bench.py --workload cfg5 --gate shut --steps 10 --warmup 2 --no-cpu-baseline > profiles/${TAG}_bench_cfg5_gate_shut.json 2>> $O/bench_err.log
timeout 600 python bench.py --workload ref_main 2>> $O/bench_err.log | tail -1 > profiles/${TAG}_bench_ref_main.json
python tools/kernel_resources.py > profiles/${TAG}_kernel_resources.txt 2>> $O/bench_err.log
# strip emulations on the final kernels (first line of stdout = the JSON; RCCL's banner follows it)
timeout 1200 python tools/strip_emulation.py --workload cfg5 --steps 10 2> $O/strips_cfg5.err | head -1 > profiles/${TAG}_strips_cfg5.json
timeout 900 python tools/strip_emulation.py --workload cfg4 --rays --gs 4 8 --steps 10 2> $O/strips_cfg4.err | head -1 > profiles/${TAG}_strips_cfg4_rays_by_ray.json
timeout 600 python tools/strip_emulation.py --workload cfg2 --rays --scene terrain --gs 8 --steps 10 2> $O/strips_terrain.err | head -1 > profiles/${TAG}_strips_cfg3_terrain.json
cp profiles/${TAG}_* $O/profiles/
rm -rf $O/*_trace $O/*_pmc_FETCH_SIZE $O/*_pmc_WRITE_SIZE $O/*_pmc_SQ_WAVES
ls -la $O/profiles | head -40
# the whole GPU suite, large maps included
timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -m gpu -q -x 2>&1 | grep -vE "^RCCL|^HIP ver|^ROCm|^Hostname|^Librccl" | tail -8 > $O/tests.txt
cat $O/tests.txt
tail -3 $O/bench_err.log
