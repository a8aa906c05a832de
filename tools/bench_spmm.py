"""Static-support SpMM: plain CSR vs blocked-union form on the renumbered benchmark graph (bf16 rows)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_outage_amd._lib as L
from multimodal_outage_amd.gwnet_engine import StaticSupport, cluster_order, _spmm
from multimodal_outage_amd.graphs import knn_graph, asym_adj
L.load()
N = 3000
A = asym_adj(knn_graph(N))
order = cluster_order([A])
sup = StaticSupport(A, 'cuda', order)
plain = StaticSupport(A[np.ix_(order, order)], 'cuda', None)
orig = StaticSupport(A, 'cuda', None)
print('max union', sup.fwd[3][3], 'mean', float(np.mean(np.diff(sup.fwd[3][1].cpu().numpy()))))
def run(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
for Tp in (12, 6, 1):
    J = 128 * Tp * 32
    X = torch.randn(N, J, device='cuda').to(torch.bfloat16)
    Yb = torch.empty(N, J, device='cuda', dtype=torch.bfloat16)
    Yf = torch.zeros(N, J, device='cuda')
    cp = run(lambda: Yb.copy_(X))
    out = [f'T={Tp} J={J} copy {cp:.0f} us']
    for name, s in (('orig-csr', orig), ('renum-csr', plain), ('blocked', sup)):
        a = run(lambda: _spmm(s.fwd, N, X, Yb, J, 0))
        b = run(lambda: _spmm(s.bwd, N, X, Yf, J, 1))
        out.append(f'{name}: bf16->bf16 {a:.0f} us, bf16->f32 beta=1 {b:.0f} us')
    print(' | '.join(out), flush=True)
