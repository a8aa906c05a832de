#!/bin/bash
# Collects the round's judged artefacts on the GPU box into gpurun_out/final/ (copy into profiles/ afterwards):
#   bench line, rocprofv3 kernel stats of the same command, FETCH_SIZE / WRITE_SIZE counter passes (separate runs).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
echo "bench done"; cut -c1-160 $O/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o run -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/under_rocprof.json 2> $O/ks.err
echo "kernel stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-unet > $O/pmc_fetch.json 2> $O/pmc_fetch.err
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-unet > $O/pmc_write.json 2> $O/pmc_write.err
echo "write done"
ls -la $O $O/ks $O/pmc_fetch | head -40
