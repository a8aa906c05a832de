"""Per-layer micro-benchmark of the UNet 3x3 convs at the config-3 shapes (134 tiles of 13x256x256): forward, data
gradient and weight gradient, fp32 arithmetic (bf16 storage) against the bf16 matrix pipe (MO_BF_MATH); HIP-event time
of un-contended launches, algorithmic bytes = every operand plane once at its stored width.
  python tools/bench_unet_conv.py [--n 134] [--only fwd,dgrad,wgrad]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_outage_amd._lib as L

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=134)
ap.add_argument('--only', default='fwd,dgrad,wgrad')
ap.add_argument('--reps', type=int, default=10)
ap.add_argument('--layers', default='', help='comma-separated indices into the layer table (default: all)')
ap.add_argument('--opt', action='append', default=[], help='name=value for mo_unet_set_option')
a = ap.parse_args()
lib = L.load()
for o in a.opt:
    k, v = o.split('=')
    L.call('mo_unet_set_option', k.encode(), int(v))
dev = 'cuda'
n, gs = a.n, 2
st = L.stream()


def timeit(fn, reps=a.reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


# (C0, bf0, C1, Co, S): first view (bf16-stored unless the network input), second view (fp32 upsampled map), size
LAYERS = [(13, 0, 0, 4, 256), (4, 1, 0, 4, 256), (4, 1, 4, 4, 256), (4, 1, 0, 8, 128), (8, 1, 0, 8, 128), (8, 1, 8, 8, 128),
          (8, 1, 0, 16, 64), (16, 1, 0, 16, 64), (16, 1, 16, 16, 64),
          # deep levels (exact-fp32 matrix pipe in both modes)
          (16, 1, 0, 32, 32), (32, 0, 0, 32, 32), (32, 0, 32, 32, 32), (32, 0, 0, 64, 16), (64, 0, 0, 64, 16)]
print(f'{"layer":26s} {"op":6s} {"fp32 us":>9s} {"bf16-mfma us":>13s} {"MB":>8s} {"TB/s (mfma)":>12s}')
if a.layers:
    LAYERS = [LAYERS[int(i)] for i in a.layers.split(',')]
for C0, bf0, C1, Co, S in LAYERS:
    Ci = C0 + C1
    G = n // gs
    x0 = torch.randn(n, C0, S, S, device=dev)
    x0 = x0.to(torch.bfloat16) if bf0 else x0
    sc = torch.rand(G, C0, device=dev) + 0.5
    sh = torch.randn(G, C0, device=dev) * 0.1
    x1 = torch.randn(n, C1, S, S, device=dev) if C1 else None
    W = torch.randn(Co, Ci, 3, 3, device=dev) / (3 * Ci ** 0.5)
    Wf = torch.empty(Ci, Co, 3, 3, device=dev)
    L.call('mo_conv3x3_flip_weights', L.ptr(W), Co, Ci, L.ptr(Wf), st)
    out = torch.empty(n, Co, S, S, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(n, Co, S, S, device=dev).to(torch.bfloat16)
    dx = torch.empty(n, Ci, S, S, device=dev, dtype=torch.bfloat16)
    dW = torch.empty(Co, Ci, 3, 3, device=dev)
    ws = torch.empty(lib.mo_unet_wgrad_ws_floats(Co, Ci * 9, n * S * S), device=dev)
    args_in = (L.ptr(x0), C0, C0 * S * S, L.ptr(sc), L.ptr(sh), 1, L.ptr(x1), C1, C1 * S * S, None, None, 0)
    name = f'{Ci:2d}->{Co:2d} @{S}^2' + (' (cat)' if C1 else '')
    px = n * S * S
    in_bytes = px * (C0 * (2 if bf0 else 4) + C1 * 4)

    def fwd(math):
        dt = L.BF_IN0 * bf0 | L.BF_OUT | (L.BF_MATH if math else 0)
        nt = lib.mo_conv3x3_stats_tiles2(C0, C1, Co, n, S, S, dt)
        stats = torch.empty(n, nt, Co, 2, device=dev)
        return lambda: L.call('mo_conv3x3_fwd', *args_in, gs, L.ptr(W), Co, n, S, S, L.ptr(out), Co * S * S, L.ptr(stats), dt, None, st)

    def dgrad(math):
        if math:
            return lambda: L.call('mo_conv3x3_fwd', L.ptr(dy), Co, Co * S * S, None, None, 0, None, 0, 0, None, None, 0, 1,
                                  L.ptr(W), Ci, n, S, S, L.ptr(dx), Ci * S * S, None,
                                  L.BF_IN0 | L.BF_OUT | L.BF_MATH | L.W_FLIP, None, st)
        return lambda: L.call('mo_conv3x3_fwd', L.ptr(dy), Co, Co * S * S, None, None, 0, None, 0, 0, None, None, 0, 1,
                              L.ptr(Wf), Ci, n, S, S, L.ptr(dx), Ci * S * S, None, L.BF_IN0 | L.BF_OUT, None, st)

    def wgrad(math):
        dt = L.BF_DY | L.BF_IN0 * bf0 | (L.BF_MATH if math else 0)
        return lambda: L.call('mo_conv3x3_bwd_weight', L.ptr(dy), Co * S * S, Co, *args_in, gs, n, S, S, L.ptr(dW), L.ptr(ws),
                              dt, None, st)

    for op, mk, nbytes in (('fwd', fwd, in_bytes + px * Co * 2), ('dgrad', dgrad, px * (Co + Ci) * 2),
                           ('wgrad', wgrad, in_bytes + px * Co * 2)):
        if op not in a.only.split(','):
            continue
        try:
            t0 = timeit(mk(0))
        except RuntimeError as e:                    # (a storage combination the fp32 kernels do not serve)
            t0 = float('nan')
        try:
            t1 = timeit(mk(1))
        except RuntimeError as e:                    # (entry point does not know MO_BF_MATH yet)
            t1 = float('nan')
        print(f'{name:26s} {op:6s} {t0:9.1f} {t1:13.1f} {nbytes / 1e6:8.1f} {nbytes / t1 / 1e6:12.2f}', flush=True)
