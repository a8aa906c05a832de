#!/bin/bash
# Instruction-mix / stall counters of the UNet conv kernels (tools/bench_unet_conv.py), three separate --pmc passes.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/conv_pmc; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU --output-format csv -d $O/p1 -o run -- python3 $R/tools/bench_unet_conv.py ${CONV_ARGS} > /dev/null 2> $O/p1.err
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/p2 -o run -- python3 $R/tools/bench_unet_conv.py ${CONV_ARGS} > /dev/null 2> $O/p2.err
rocprofv3 --kernel-trace --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM --output-format csv -d $O/p3 -o run -- python3 $R/tools/bench_unet_conv.py ${CONV_ARGS} > /dev/null 2> $O/p3.err
ls $O/p1 $O/p2 $O/p3
