O=$GRAFT_REPO_ROOT/gpurun_out/r02f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_BRANCH" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT" "SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_LEVEL_WAVES SQ_INSTS_VALU_CVT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $O/pmc_$i -- python $R/bench.py --workload cfg3 --steps 12 --warmup 2 --no-cpu-baseline --no-cfg3 > $O/pmc_$i.log 2>&1
done
rocprofv3 --kernel-trace --stats -d $O/trace -- python $R/bench.py --workload cfg3 --steps 12 --warmup 2 --no-cpu-baseline > $O/trace.log 2>&1
ls $O
