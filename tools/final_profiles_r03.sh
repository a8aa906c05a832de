#!/bin/bash
# Collects the round's judged artefacts (round 3) on the GPU box into gpurun_out/final_r03/ (copy into profiles/ afterwards):
#   bench line, rocprofv3 kernel stats / trace of the same command, FETCH_SIZE / WRITE_SIZE counter passes (separate
#   runs, kernel-trace only beside them), a UNet-only kernel-stats run, and the kernel trace of the alternative stream
#   structure (dense data-path products on a side stream) for the timeline comparison.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final_r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
echo "bench done"; cut -c1-160 $O/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o run -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/under_rocprof.json 2> $O/ks.err
echo "kernel stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/unet_ks -o run -- python3 $R/tools/bench_unet.py --batch 1 --horizon 2 --size 256 --cin 13 --steps 10 --warmup 3 > $O/unet_under_rocprof.json 2> $O/unet_ks.err
echo "unet kernel stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-unet > $O/pmc_fetch.json 2> $O/pmc_fetch.err
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-unet > $O/pmc_write.json 2> $O/pmc_write.err
echo "write done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/unet_pmc_fetch -o run -- python3 $R/tools/bench_unet.py --batch 1 --horizon 2 --size 256 --cin 13 --steps 3 --warmup 1 > $O/unet_pmc_fetch.json 2> $O/unet_pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/unet_pmc_write -o run -- python3 $R/tools/bench_unet.py --batch 1 --horizon 2 --size 256 --cin 13 --steps 3 --warmup 1 > $O/unet_pmc_write.json 2> $O/unet_pmc_write.err
echo "unet pmc done"
python3 $R/tools/bench_unet_conv.py > $O/unet_conv_layers.txt 2> $O/unet_conv_layers.err
for b in 1 2 4 8; do python3 $R/tools/bench_unet.py --batch $b --horizon 2 --size 256 --cin 13 --steps 10 --warmup 3 2>/dev/null | tail -1; done > $O/unet_batch_sweep.txt
echo "unet layers + batch sweep done"
ls -la $O | head -40
