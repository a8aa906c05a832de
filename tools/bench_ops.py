"""Op-level micro-benchmark of the HBM-bound gwnet kernels at the bench shape (B windows of N=3000, layer 0:
Tin=13 -> Tout=12, 7 sources), un-contended: algorithmic bytes / HIP-event time per launch."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_outage_amd._lib as L
from multimodal_outage_amd.graphs import knn_graph, csr_from_dense, asym_adj

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--tout', type=int, default=12)
ap.add_argument('--only', default='')
ap.add_argument('--mf', type=int, default=0, help='1: bf16 MFMA in the TCN data path')
a = ap.parse_args()
lib = L.load()
MF = a.mf
N, B, Tout = 3000, a.batch, a.tout
Tin = Tout + 1
G = N * B
P = G * Tout
J = B * Tout * 32
ns = 7
st = L.stream()
dev = 'cuda'


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def report(name, ms, nbytes):
    print(f'{name:34s} {ms * 1e3:8.1f} us  {nbytes / 1e6:9.1f} MB algorithmic  {nbytes / ms / 1e9:6.2f} TB/s '
          f'({nbytes / ms / 1e9 / 8.0 * 100:4.1f}% of 8 TB/s)', flush=True)


def want(k):
    return (not a.only) or (k in a.only.split(','))


row = P * 32 * 4       # one [P][32] fp32 tensor
srcs = [torch.randn(P, 32, device=dev) for _ in range(ns)]
W = torch.randn(32, 32 * ns, device=dev) / np.sqrt(32 * ns)
b = torch.randn(32, device=dev)
res = torch.randn(G * Tin, 32, device=dev)
sc = torch.rand(32, device=dev) + 0.5
sh = torch.randn(32, device=dev)
h = torch.empty(P, 32, device=dev)
partial = torch.empty(lib.mo_mlp_partial_floats(P), device=dev)
thresh, dscale = int(0.3 * 4294967296.0), 1.0 / 0.7

if want('spmm'):
    rs = np.random.RandomState(0)
    A = knn_graph(rs.uniform(size=(N, 2)).astype(np.float32), 6) if False else None
    # the bench's supports: symmetric k-NN graph over random 2-D positions, row-normalised
    pos = rs.uniform(size=(N, 2)).astype(np.float32)
    d2 = ((pos[:, None, :] - pos[None, :, :]) ** 2).sum(-1)
    idx = np.argsort(d2, axis=1)[:, :6]
    A = np.zeros((N, N), np.float32)
    for i in range(N):
        A[i, idx[i]] = 1.0
    A = A / A.sum(1, keepdims=True)
    rp, ci, va = csr_from_dense(A.T)
    csr = [torch.from_numpy(x).to(dev) for x in (rp, ci, va)]
    X, Y = srcs[0], srcs[1]
    ms = timeit(lambda: L.call('mo_spmm_csr', L.ptr(csr[0]), L.ptr(csr[1]), L.ptr(csr[2]), N, L.ptr(X), L.ptr(Y), J,
                               0, 0, 0, st))
    report(f'spmm_csr beta=0 (nnz {len(va)})', ms, 2 * row)
    ms = timeit(lambda: L.call('mo_spmm_csr', L.ptr(csr[0]), L.ptr(csr[1]), L.ptr(csr[2]), N, L.ptr(X), L.ptr(Y), J,
                               1, 0, 0, st))
    report('spmm_csr beta=1', ms, 3 * row)
    rpI, ciI, vaI = csr_from_dense(np.eye(N, dtype=np.float32))
    csrI = [torch.from_numpy(x).to(dev) for x in (rpI, ciI, vaI)]
    ms = timeit(lambda: L.call('mo_spmm_csr', L.ptr(csrI[0]), L.ptr(csrI[1]), L.ptr(csrI[2]), N, L.ptr(X), L.ptr(Y), J,
                               0, 0, 0, st))
    report('spmm_csr identity (pure copy)', ms, 2 * row)
    ms = timeit(lambda: Y.copy_(X))
    report('torch copy_', ms, 2 * row)

if want('mlp_fwd'):
    ms = timeit(lambda: L.call('mo_gcn_mlp_fwd', L.ptr_array(srcs), ns, L.ptr(W), L.ptr(b), G, Tout, Tin,
                               L.ptr(res), L.ptr(sc), L.ptr(sh), 1, thresh, dscale, L.ptr(h), L.ptr(partial), 0, st))
    report('gcn_mlp_fwd (7 src + res -> h)', ms, (ns + 2) * row)

dh = torch.randn(P, 32, device=dev)
dsrcs = [torch.empty(P, 32, device=dev) for _ in range(ns)]
dW = torch.empty(32, 32 * ns, device=dev); db = torch.empty(32, device=dev)
wsm = torch.empty(lib.mo_wgrad_ws_floats(32, 32 * ns, P), device=dev)
dlast = torch.empty(P, 32, device=dev, dtype=torch.bfloat16)
if want('mlp_bwd'):
    ms = timeit(lambda: L.call('mo_gcn_mlp_bwd', L.ptr(dh), L.ptr_array(srcs), L.ptr_array(dsrcs), ns, L.ptr(W), P,
                               1, thresh, dscale, L.ptr(dW), L.ptr(db), None, L.ptr(dlast), 1, 0, 0, st))
    report('gcn_mlp_bwd data (dh -> 7 dsrc)', ms, (ns + 1) * row + row // 2)
    ms = timeit(lambda: L.call('mo_gcn_mlp_bwd', L.ptr(dh), L.ptr_array(srcs), L.ptr_array(dsrcs), ns, L.ptr(W), P,
                               1, thresh, dscale, L.ptr(dW), L.ptr(db), L.ptr(wsm), None, 2, 0, 0, st))
    report('gcn_mlp_bwd wgrad (dh, 7 src)', ms, (ns + 1) * row)

if want('tcn'):
    K, d = 2, 1
    Wf = torch.randn(32, 32, 1, K, device=dev) * 0.1
    Wg = torch.randn(32, 32, 1, K, device=dev) * 0.1
    Wp = torch.empty(K * 64 * 32, device=dev)
    L.call('mo_tcn_pack_weights', L.ptr(Wf), L.ptr(Wg), K, L.ptr(Wp), st)
    g = torch.empty(P, 32, device=dev)
    gbf = torch.empty(P, 32, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: L.call('mo_tcn_fwd', L.ptr(res), L.ptr(sc), L.ptr(sh), L.ptr(Wp), L.ptr(b), L.ptr(b), K, d, G,
                               Tin, L.ptr(g), L.ptr(gbf), MF, None, 0, st))
    report('tcn_fwd (h -> g, g_bf16)', ms, int(row * (Tin / Tout + 1.5)))
    du = torch.empty(G * Tin, 32, device=dev)
    dpre = torch.empty(P * 64, device=dev)
    ws2 = torch.empty(lib.mo_wgrad_ws_floats(64, 32 * K, P), device=dev)
    gW = [torch.empty(32, 32, 1, K, device=dev) for _ in range(2)]
    gb = [torch.empty(32, device=dev) for _ in range(2)]
    ms = timeit(lambda: L.call('mo_tcn_bwd', L.ptr(res), L.ptr(sc), L.ptr(sh), L.ptr(Wp), L.ptr(b), L.ptr(b), K, d, G,
                               Tin, L.ptr(dh), L.ptr(dsrcs[0]), L.ptr(du), L.ptr(gW[0]), L.ptr(gW[1]), L.ptr(gb[0]),
                               L.ptr(gb[1]), L.ptr(dpre), L.ptr(ws2), 1, MF, st))
    report('tcn_bwd data (dg,h -> dpre,du)', ms, int(row * (Tin / Tout * 2 + 1 + 2 + 1)))
    ms = timeit(lambda: L.call('mo_tcn_bwd', L.ptr(res), L.ptr(sc), L.ptr(sh), L.ptr(Wp), L.ptr(b), L.ptr(b), K, d, G,
                               Tin, L.ptr(dh), L.ptr(dsrcs[0]), L.ptr(du), L.ptr(gW[0]), L.ptr(gW[1]), L.ptr(gb[0]),
                               L.ptr(gb[1]), L.ptr(dpre), L.ptr(ws2), 2, MF, st))
    report('tcn_bwd wgrad (dpre, h)', ms, int(row * (2 + Tin / Tout)))

if want('bn'):
    stats = torch.rand(4, 32, device=dev) + 0.5
    wsb = torch.empty(lib.mo_mlp_partial_floats(P) + 64, device=dev)
    gg = torch.empty(32, device=dev); gb_ = torch.empty(32, device=dev)
    ms = timeit(lambda: L.call('mo_bn_bwd', L.ptr(dh), L.ptr(h), P, L.ptr(sc), L.ptr(stats[2]), L.ptr(stats[3]),
                               L.ptr(srcs[0]), L.ptr(gg), L.ptr(gb_), L.ptr(wsb), st))
    report('bn_bwd (2 passes)', ms, 5 * row)

if want('skip'):
    Cs = 256
    Tf = 1
    Pf = G * Tf
    Ws = torch.randn(Cs, 32, device=dev) * 0.1
    bs = torch.randn(Cs, device=dev)
    skip = torch.empty(Pf, Cs, device=dev)
    g = srcs[0]
    ms = timeit(lambda: L.call('mo_conv1x1_fwd', L.ptr(g), 32, Tf, Tout, Tout - Tf, 0, L.ptr(Ws), L.ptr(bs), Cs,
                               L.ptr(skip), Pf, 0, 1, st))
    report('skip conv fwd (beta=1)', ms, Pf * (32 + 2 * Cs) * 4)
    dWs = torch.empty(Cs, 32, device=dev)
    wss = torch.empty(lib.mo_wgrad_ws_floats(Cs, 32, Pf), device=dev)
    ms = timeit(lambda: L.call('mo_conv1x1_bwd_weight', L.ptr(skip), Cs, Pf, L.ptr(g), 32, Tf, Tout, Tout - Tf, 0,
                               L.ptr(dWs), None, L.ptr(wss), st))
    report('skip conv wgrad', ms, Pf * (32 + Cs) * 4)
    ms = timeit(lambda: L.call('mo_conv1x1_bwd_data', L.ptr(skip), Cs, Pf, L.ptr(Ws), 32, L.ptr(dsrcs[0]), Tf, Tout,
                               Tout - Tf, None, 1, st))
    report('skip conv bwd data (beta=1)', ms, Pf * (Cs + 64) * 4)

if want('bf16'):
    # the same ops with the diffusion intermediates stored as bf16 (throughput mode)
    srcs_b = [srcs[0]] + [s_.to(torch.bfloat16) for s_ in srcs[1:]]
    mask = ((1 << ns) - 1) & ~1
    ms = timeit(lambda: L.call('mo_gcn_mlp_fwd', L.ptr_array(srcs_b), ns, L.ptr(W), L.ptr(b), G, Tout, Tin,
                               L.ptr(res), L.ptr(sc), L.ptr(sh), 1, thresh, dscale, L.ptr(h), L.ptr(partial), mask, st))
    report('bf16-stored: gcn_mlp_fwd', ms, (1 + (ns - 1) / 2 + 2) * row)
    dsrcs_b = [dsrcs[0]] + [torch.empty(P, 32, device=dev, dtype=torch.bfloat16) for _ in range(ns - 1)]
    ms = timeit(lambda: L.call('mo_gcn_mlp_bwd', L.ptr(dh), L.ptr_array(srcs_b), L.ptr_array(dsrcs_b), ns, L.ptr(W), P,
                               1, thresh, dscale, L.ptr(dW), L.ptr(db), None, None, 1, mask, mask, st))
    report('bf16-stored: gcn_mlp_bwd data', ms, (2 + (ns - 1) / 2) * row)
    ms = timeit(lambda: L.call('mo_gcn_mlp_bwd', L.ptr(dh), L.ptr_array(srcs_b), L.ptr_array(dsrcs_b), ns, L.ptr(W), P,
                               1, thresh, dscale, L.ptr(dW), L.ptr(db), L.ptr(wsm), None, 2, mask, mask, st))
    report('bf16-stored: gcn_mlp_bwd wgrad', ms, (2 + (ns - 1) / 2) * row)
    Xb, Yb = srcs_b[1], srcs_b[2]
    rs = np.random.RandomState(0)
    pos = rs.uniform(size=(N, 2)).astype(np.float32)
    d2 = ((pos[:, None, :] - pos[None, :, :]) ** 2).sum(-1)
    idx = np.argsort(d2, axis=1)[:, :6]
    A = np.zeros((N, N), np.float32)
    for i in range(N):
        A[i, idx[i]] = 1.0
    A = A / A.sum(1, keepdims=True)
    csr = [torch.from_numpy(x).to(dev) for x in csr_from_dense(A.T)]
    for (xx, yy, nm, nb) in ((srcs[0], Yb, 'f32->bf16', 1.5), (Xb, Yb, 'bf16->bf16', 1.0), (Xb, srcs[1], 'bf16->f32 beta=1', 2.5)):
        ms = timeit(lambda: L.call('mo_spmm_csr', L.ptr(csr[0]), L.ptr(csr[1]), L.ptr(csr[2]), N, L.ptr(xx), L.ptr(yy), J,
                                   1 if 'beta' in nm else 0, int(xx.dtype == torch.bfloat16),
                                   int(yy.dtype == torch.bfloat16), st))
        report(f'bf16-stored: spmm {nm}', ms, nb * row)
