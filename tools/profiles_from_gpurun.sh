#!/bin/bash
# local: convert gpurun_out/prof_<tag>/ (written by tools/profile_round.sh on the GPU box) into committed summaries.
TAG=${1:-r01}
O=gpurun_out/prof_$TAG
for wl in cfg2 cfg3; do
  (echo "# $TAG: rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl (see tools/profile_round.sh)"; python tools/rocprof_summary.py $(ls $O/${wl}_trace/*/*_results.db | head -1)) > profiles/${TAG}_${wl}_kernel_stats.txt
  python tools/pmc_to_json.py $(ls $O/${wl}_pmc_FETCH_SIZE/*/*_results.db | head -1) $(ls $O/${wl}_pmc_WRITE_SIZE/*/*_results.db | head -1) > profiles/${TAG}_pmc_${wl}.json
  cp profiles/${TAG}_pmc_${wl}.json profiles/pmc_${wl}.json
  python - <<PY > profiles/${TAG}_${wl}_sq_counters.txt
import sqlite3,glob
f=glob.glob("$O/${wl}_pmc_SQ_WAVES/*/*_results.db")[0]
cur=sqlite3.connect(f).cursor()
rows=cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
print("# $TAG $wl: rocprofv3 --pmc SQ_* (average per dispatch)")
for r in rows: print("%-64s %-22s %16.1f n=%d"%(r[0][:64],r[1],r[2],r[3]))
PY
done
ls -la profiles/
