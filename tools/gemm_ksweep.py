"""Per-tile fixed cost of the ring GEMM: time(K) = a + b*K at fixed M=3000, J (bf16 result), so that a/(a+3000b) is
what a persistent / overlapped prologue-epilogue form could at most recover on the K=3000 products."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_outage_amd._lib as L
L.load()
M = 3000
st = L.stream()
for J in (24576, 98304):
    res = []
    for K in (512, 1504, 3008, 6016):
        A = torch.randn(M, K, device='cuda').to(torch.bfloat16)
        X = torch.randn(K, J, device='cuda').to(torch.bfloat16)
        Yb = torch.empty(M, J, device='cuda', dtype=torch.bfloat16)
        def run():
            L.call('mo_gemm_bf16_256', L.ptr(A), K, K, L.ptr(X), J, 1, None, J, M, J, K, 0, L.ptr(Yb), st)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10
        res.append((K, t))
        print(f'J={J} K={K}: {t*1e3:.0f} us  {2.0*M*J*K/t/1e9:.0f} TF', flush=True)
        del A, X, Yb
    (k1, t1), (k2, t2) = res[1], res[3]
    b = (t2 - t1) / (k2 - k1); a = t1 - b * k1
    print(f'J={J}: fixed {a*1e3:.0f} us + {b*1e3:.4f} us/k  -> at K=3008 fixed share {a/(a+3008*b):.3f}; asymptotic {2.0*M*J/b/1e9:.0f} TF')
