#!/bin/bash
# Instruction-mix / stall counters of the blocked SpMM (tools/bench_spmm.py), three separate --pmc passes.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/spmm_pmc; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $O/p1 -o run -- python3 $R/tools/bench_spmm.py > $O/p1.out 2> $O/p1.err
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/p2 -o run -- python3 $R/tools/bench_spmm.py > /dev/null 2> $O/p2.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES --output-format csv -d $O/p3 -o run -- python3 $R/tools/bench_spmm.py > /dev/null 2> $O/p3.err
cat $O/p1.out | tail -4
