set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_unet_ks -o run -- python3 $R/tools/bench_unet.py --batch ${UB:-1} --horizon 2 --size 256 --cin 13 --steps 10 --warmup 3 > $R/gpurun_out/r3_unet_ks.json 2> $R/gpurun_out/r3_unet_ks.err
echo done
