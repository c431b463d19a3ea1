O=$GRAFT_REPO_ROOT/gpurun_out/r02e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -E "^\s*(Name|name)?.*SQ_" | head -150 > $O/avail_sq.txt
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z0-9_]+" | sort -u > $O/sq_names.txt
wc -l $O/sq_names.txt
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_BRANCH" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_CVT SQ_VALU_MFMA_BUSY_CYCLES"; do
  n=$(echo $set | cut -d" " -f1)
  rocprofv3 --kernel-trace --pmc $set -d $O/pmc_$n -- python $R/bench.py --workload cfg3 --steps 4 --warmup 1 --no-cpu-baseline > $O/pmc_$n.log 2>&1
  tail -2 $O/pmc_$n.log | cut -c1-300
done
find $O -name "*.csv" | head; find $O -name "*.db" | head
