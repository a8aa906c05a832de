set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_unet_ops_gpu.py -x -q -m gpu -k "conv3x3" > gpurun_out/ux_t.log 2>&1 || { tail -30 gpurun_out/ux_t.log; exit 1; }
tail -2 gpurun_out/ux_t.log
python tools/bench_unet_conv.py --layers 1,2,3 2>&1 | grep "fwd\|dgrad"
for b in 1 4 1 4; do python tools/bench_unet.py --batch $b --horizon 2 --size 256 --cin 13 --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-120; sleep 10; done
