# GPU box: rocprofv3 kernel-trace stats of one bench workload -> gpurun_out/<tag>/<wl>_stats.txt.  usage: tools/prof_quick.sh <tag> <workload> [bench args]
TAG=${1:-pq}; WL=${2:-cfg2}; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/${WL}_trace -- python $R/bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline --no-cfg3 "$@" > $O/${WL}_trace.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/${WL}_trace/*/*_results.db | head -1) > $O/${WL}_stats.txt
head -14 $O/${WL}_stats.txt | cut -c1-150
