"""Host enqueue time vs GPU time per training step (is the step host-bound?)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_outage_amd._lib as L
L.load()
from multimodal_outage_amd.models.graph_wavenet import gwnet
from multimodal_outage_amd.trainer import FlatTrainer
from multimodal_outage_amd.graphs import knn_graph, asym_adj
N = 3000
A = knn_graph(N)
sup = [asym_adj(A), asym_adj(A.T)]
torch.manual_seed(42)
m = gwnet('cpu', num_nodes=N, dropout=0.3, supports=sup, in_dim=32, out_dim=12, kernel_size=2).cuda().train()
m.dense_dtype = 'bf16'
tr = FlatTrainer(m)
m._mo_grad_out = tr.grad_out()
for B in (16, 32, 64):
    x = torch.randn(B, 32, N, 12, device='cuda'); y = torch.randn(B, 12, N, 1, device='cuda')
    n = y.numel(); sums = torch.empty(4, device='cuda'); dy = torch.empty_like(y)
    ws = torch.empty(L.load().mo_metrics_ws_floats(n), device='cuda')
    def step():
        out = m(x)
        L.call('mo_mse_metrics', L.ptr(out), L.ptr(y), n, L.ptr(sums), L.ptr(dy), L.ptr(ws), L.stream())
        out.backward(dy); tr.step()
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): step()
    t_enq = (time.perf_counter() - t0) / 5
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 5
    print(f'B={B}: host enqueue {t_enq*1e3:.1f} ms/step, wall {t_all*1e3:.1f} ms/step')
