#!/bin/bash
# hardware counters of the grouped weight-gradient kernel on the cls model's queue (tools/dw_bench.py), one --pmc pass per group
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/dw_counters
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o dw -- python3 $R/tools/dw_bench.py > $OUT/p$i.log 2>&1
  python3 $R/tools/pmc_summary.py $(ls $OUT/p$i/*counter_collection.csv | head -1) 60 | grep -E "gemm_tn_grouped|splitk_reduce_grouped" >> $OUT/summary.txt
  rm -rf $OUT/p$i
done
cat $OUT/summary.txt
