"""Yardstick only (never on the product path): what the vendor bf16 GEMM (torch.matmul -> hipBLASLt/rocBLAS) reaches on
the shapes of the dense node-axis products, beside the hand-written ring kernel (tools/bench_gemm.py)."""
import torch
N = 3000
A = torch.randn(N, N, device='cuda').to(torch.bfloat16)
for J in (3072, 6144, 12288, 24576, 49152):
    X = torch.randn(N, J, device='cuda').to(torch.bfloat16)
    def run(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10
    fl = 2.0 * N * N * J
    t1 = run(lambda: torch.matmul(A, X))
    t2 = run(lambda: torch.matmul(X.t().contiguous().t() if False else X, X.t()))     # dA-shaped: [N,J] x [J,N]
    print(f'J={J:6d} product {fl/t1/1e9:7.1f} TF ({t1*1e3:.0f} us)   dA-shaped {fl/t2/1e9:7.1f} TF ({t2*1e3:.0f} us)', flush=True)
