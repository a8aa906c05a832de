"""What the first conv (13 fp32 channels -> 4, bf16 matrix pipe) would cost without its halo re-fetch: the same pixel count
as 134 x 256 x 256 in shapes with fewer tile neighbours.   python tools/exp_first_conv.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_outage_amd._lib as L
lib = L.load()
dev = 'cuda'
st = L.stream()


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


import sys as _s
if len(_s.argv) > 1:
    L.call('mo_unet_set_option', b'ub_no_pack', int(_s.argv[1]))
for n, H, W in ((134, 256, 256), (134, 1024, 64), (134, 64, 1024), (134 * 4, 128, 128), (134 * 16, 64, 64)):
    C0, Co, gs = 13, 4, 2
    x0 = torch.randn(n, C0, H, W, device=dev)
    Wt = torch.randn(Co, C0, 3, 3, device=dev) / 10
    out = torch.empty(n, Co, H, W, device=dev, dtype=torch.bfloat16)
    dt = L.BF_OUT | L.BF_MATH
    nt = lib.mo_conv3x3_stats_tiles2(C0, 0, Co, n, H, W, dt)
    stats = torch.empty(n, max(nt, 1), Co, 2, device=dev)
    t = timeit(lambda: L.call('mo_conv3x3_fwd', L.ptr(x0), C0, C0 * H * W, None, None, 0, None, 0, 0, None, None, 0, gs,
                              L.ptr(Wt), Co, n, H, W, L.ptr(out), Co * H * W, L.ptr(stats), dt, None, st))
    mb = n * H * W * (C0 * 4 + Co * 2) / 1e6
    print(f'{n:5d} x {C0} x {H:4d} x {W:4d}: {t:7.1f} us  {mb / t:6.2f} TB/s of {mb:.0f} MB', flush=True)
