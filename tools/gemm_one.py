"""One shape of the ring GEMM (forward product, bf16 result, and the dA shape) for counter collection:
rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d out -- python3 tools/gemm_one.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_outage_amd._lib as L
L.load()
N, kpad, J = 3000, 3008, 24576
A = torch.randn(N, kpad, device='cuda').to(torch.bfloat16); A[:, N:] = 0
X = torch.randn(N, J, device='cuda').to(torch.bfloat16)
Yb = torch.empty(N, J, device='cuda', dtype=torch.bfloat16)
dA = torch.zeros(N, N, device='cuda')
st = L.stream()
for _ in range(4):
    L.call('mo_gemm_bf16_256', L.ptr(A), kpad, kpad, L.ptr(X), J, 1, None, J, N, J, N, 0, L.ptr(Yb), st)
    L.call('mo_gemm_bf16_256', L.ptr(X), J, J, L.ptr(X), J, 0, L.ptr(dA), N, N, N, J, 1, None, st)
torch.cuda.synchronize()
