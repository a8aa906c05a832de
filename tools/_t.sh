cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/loss_ks -o run -- python3 $R/tools/bench_unet.py --batch 1 --horizon 2 --size 256 --cin 13 --steps 10 --warmup 3 > /dev/null 2>&1
grep "outc_loss" $R/gpurun_out/loss_ks/run_kernel_stats.csv | cut -c1-120
