#!/bin/bash
# LDS bank-conflict / wait counters of every kernel of the gwnet step (one --pmc pass beside a kernel trace); read with
#   python3 tools/unet_lds_counters.py gpurun_out/gwnet_lds/run_counter_collection.csv
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/gwnet_lds; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d $O -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-unet > $O/out.json 2> $O/err.txt
ls $O
