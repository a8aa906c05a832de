"""What the adjacency gradient would cost with k-major operands: dA = Xt^T dYt with Xt, dYt = [J][N] bf16 (the
split-K kk ring kernel) against today's [N][J] x [N][J] product, plus the cost of a bf16 transpose of one operand."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_outage_amd._lib as L
lib = L.load()
N, Np = 3000, 3008          # k-major rows padded to a multiple of 8 columns
st = L.stream()
def run(f):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 8
for J in (24576, 49152, 98304):
    X = torch.randn(N, J, device='cuda').to(torch.bfloat16)
    dY = torch.randn(N, J, device='cuda').to(torch.bfloat16)
    Xt = torch.zeros(J, Np, device='cuda', dtype=torch.bfloat16); Xt[:, :N] = X.t()
    dYt = torch.zeros(J, Np, device='cuda', dtype=torch.bfloat16); dYt[:, :N] = dY.t()
    dA = torch.zeros(N, N, device='cuda')
    dA2 = torch.empty(Np, Np, device='cuda')
    ws = torch.empty(lib.mo_wgrad_bf16_kk_ws_floats(Np, Np, J), device='cuda')
    fl = 2.0 * N * N * J
    t0 = run(lambda: L.call('mo_gemm_bf16_256', L.ptr(X), J, J, L.ptr(dY), J, 0, L.ptr(dA), N, N, N, J, 1, None, st))
    t1 = run(lambda: L.call('mo_wgrad_bf16_kk', L.ptr(Xt), Np, L.ptr(dYt), Np, J, Np, Np, L.ptr(dA2), L.ptr(ws), st))
    tt = run(lambda: Xt[:, :N].copy_(X.t()))
    L.call('mo_gemm_bf16_256', L.ptr(X), J, J, L.ptr(dY), J, 0, L.ptr(dA), N, N, N, J, 0, None, st)
    torch.cuda.synchronize()
    err = float((dA2[:N, :N] - dA).abs().max()) / float(dA.abs().max())
    print(f'J={J:6d}  [N][J] operands {t0*1e3:6.0f} us ({fl/t0/1e9:5.0f} TF) | k-major operands {t1*1e3:6.0f} us ({fl/t1/1e9:5.0f} TF) '
          f'| torch transpose of one operand {tt*1e3:5.0f} us | rel diff {err:.1e}', flush=True)
