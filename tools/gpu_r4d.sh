O=gpurun_out/r4d; mkdir -p $O
for rep in 1 2; do for f in tools/ab/a_old.so tools/ab/b_new.so; do
  EMAP_HIP_LIB=$PWD/$f timeout 300 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline --interleaved-cloud 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
st=r['stage_ms']; tot=sum(st.values())
print('$f', '%.4f ms' % d['ms_per_step'], 'semantic ~ %.3f' % (d['ms_per_step']-tot+0.03), {k: round(v*1e3,1) for k,v in st.items() if v > 0.01})"
done; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o cfg5 -- python $GRAFT_REPO_ROOT/bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats*" | head; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 $f | cut -c1-200
