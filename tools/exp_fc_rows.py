"""FC 3 x bf16 kernels at several row counts (windows per step x 134), one vs two 16-column blocks of W per wave.
  python tools/exp_fc_rows.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_outage_amd._lib as L
lib = L.load()
dev = 'cuda'


def t(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for P in (134, 536, 1072):
    for (Co, Ci) in ((4096, 16384), (16384, 1024)):
        W = torch.randn(Co, Ci, device=dev) * 0.01; b = torch.randn(Co, device=dev)
        x = torch.randn(P, Ci, device=dev); dout = torch.randn(P, Co, device=dev)
        out = torch.empty(P, Co, device=dev); din = torch.empty(P, Ci, device=dev)
        ws3 = torch.empty(max(lib.mo_fc3_ws_floats(P, Ci, Co), lib.mo_fc3_ws_floats(P, Co, Ci)), device=dev)
        r = []
        for mode in (0, 1):
            L.call('mo_unet_set_option', b'fc_wide', mode)
            f3 = t(lambda: L.call('mo_fc3_fwd', L.ptr(x), P, Ci, L.ptr(W), L.ptr(b), Co, 1, L.ptr(out), L.ptr(ws3), L.stream()))
            d3 = t(lambda: L.call('mo_fc3_bwd_data', L.ptr(dout), P, Co, L.ptr(W), Ci, L.ptr(din), L.ptr(ws3), L.stream()))
            r.append((f3, d3))
        L.call('mo_unet_set_option', b'fc_wide', 1)
        print(f'P {P:5d} W {Co}x{Ci}: one column block per wave: fwd {r[0][0]:7.1f} us dgrad {r[0][1]:7.1f} us | two: fwd {r[1][0]:7.1f} dgrad {r[1][1]:7.1f}', flush=True)
