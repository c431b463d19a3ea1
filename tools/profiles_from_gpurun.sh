#!/bin/bash
# local: convert gpurun_out/prof_<tag>/ (written by tools/profile_round.sh on the GPU box) into committed summaries.  Every file is
# stamped with the sha256 of the kernel sources the profiled library was built from (bench.py: source_stamp); the script REFUSES to
# convert a profile whose stamp (written on the GPU box next to the databases) differs from the current sources -- profiles of one
# binary must not sit next to timings of another.
TAG=${1:-r01}
O=gpurun_out/prof_$TAG
NOW=$(python -c "import bench; print(bench.source_stamp())")
WAS=$(cat $O/source_stamp.txt 2>/dev/null)
if [ "$NOW" != "$WAS" ]; then echo "REFUSED: $O was profiled with kernel sources $WAS, the tree now has $NOW -- re-run tools/profile_round.sh"; exit 1; fi
for wl in cfg2 cfg3; do
  (echo "# $TAG: rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl (see tools/profile_round.sh)"; echo "# source_stamp: $NOW"; python tools/rocprof_summary.py $(ls $O/${wl}_trace/*/*_results.db | head -1)) > profiles/${TAG}_${wl}_kernel_stats.txt
  python tools/pmc_to_json.py $(ls $O/${wl}_pmc_FETCH_SIZE/*/*_results.db | head -1) $(ls $O/${wl}_pmc_WRITE_SIZE/*/*_results.db | head -1) $NOW > profiles/${TAG}_pmc_${wl}.json
  python - <<PY > profiles/${TAG}_${wl}_sq_counters.txt
import sqlite3,glob
f=glob.glob("$O/${wl}_pmc_SQ_WAVES/*/*_results.db")[0]
cur=sqlite3.connect(f).cursor()
rows=cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
print("# $TAG $wl: rocprofv3 --pmc SQ_* (average per dispatch)")
print("# source_stamp: $NOW")
for r in rows: print("%-64s %-22s %16.1f n=%d"%(r[0][:64],r[1],r[2],r[3]))
PY
done
for f in bench_cfg1 bench_cfg2 bench_cfg2_shifted bench_cfg3 bench_cfg4 bench_cfg5; do [ -s $O/$f.json ] && cp $O/$f.json profiles/${TAG}_$f.json; done
ls -la profiles/ | grep $TAG
