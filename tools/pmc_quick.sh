# GPU box: rocprofv3 kernel stats + HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes) of one bench workload -> gpurun_out/<tag>/.
# usage: tools/pmc_quick.sh <tag> <workload> [bench args]
TAG=${1:-pq}; WL=${2:-cfg2}; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/${WL}_trace -- python $R/bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline --no-cfg3 --no-large "$@" > $O/${WL}_trace.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/${WL}_trace/*/*_results.db | head -1) > $O/${WL}_stats.txt
head -14 $O/${WL}_stats.txt | cut -c1-150
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/${WL}_pmc_$c -- python $R/bench.py --workload $WL --steps 4 --warmup 2 --no-cpu-baseline --no-cfg3 --no-large "$@" > $O/${WL}_pmc_$c.log 2>&1
done
python $R/tools/pmc_to_json.py $(ls $O/${WL}_pmc_FETCH_SIZE/*/*_results.db | head -1) $(ls $O/${WL}_pmc_WRITE_SIZE/*/*_results.db | head -1) > $O/${WL}_pmc.json
python - <<PY
import json
d=json.load(open("$O/${WL}_pmc.json"))
tot=0
for k,v in d['kernels'].items():
    if v['hbm_bytes']>1e6: print("%-44s read %8.1f MB  write %8.1f MB  n=%d"%(k[:44],v['hbm_read_bytes']/1e6,v['hbm_write_bytes']/1e6,v['launches']))
PY
rm -rf $O/${WL}_trace $O/${WL}_pmc_FETCH_SIZE $O/${WL}_pmc_WRITE_SIZE
