#!/bin/bash
# Counter passes over tools/gemm_one.py (each pass on its own, kernel trace only) -> gpurun_out/gemm_pmc/passN
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/gemm_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
         "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pass$i -o run -- python3 $R/tools/gemm_one.py > $O/pass$i.log 2>&1 || { tail -5 $O/pass$i.log; echo "pass $i failed"; }
  echo "pass $i done"
done
