"""Micro-benchmark of the dense node-axis product kernels (one process, interleaved rounds)."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_outage_amd._lib as L
L.load()
N = 3000
kpad = 3008
A = torch.randn(N, kpad, device='cuda').to(torch.bfloat16); A[:, N:] = 0
for J in (1024, 3072, 6144, 12288, 24576):
    X = torch.randn(N, J, device='cuda').to(torch.bfloat16)
    Y = torch.empty(N, J, device='cuda')
    dA = torch.empty(N, N, device='cuda')
    st = L.stream()
    def run(name, *a):
        for _ in range(2): L.call(name, *a)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): L.call(name, *a)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10
    fl = 2.0 * N * N * J
    t128 = run('mo_gemm_bf16', L.ptr(A), kpad, L.ptr(X), J, 1, L.ptr(Y), J, N, J, N, 0, None, st)
    t256 = run('mo_gemm_bf16_256', L.ptr(A), kpad, kpad, L.ptr(X), J, 1, L.ptr(Y), J, N, J, N, 0, None, st)
    Yb = torch.empty(N, J, device='cuda', dtype=torch.bfloat16)
    tb0 = run('mo_gemm_bf16_256', L.ptr(A), kpad, kpad, L.ptr(X), J, 1, None, J, N, J, N, 0, L.ptr(Yb), st)
    tb1 = run('mo_gemm_bf16_256', L.ptr(A), kpad, kpad, L.ptr(X), J, 1, None, J, N, J, N, 1, L.ptr(Yb), st)
    g128 = run('mo_gemm_bf16', L.ptr(X), J, L.ptr(X), J, 0, L.ptr(dA), N, N, N, J, 0, None, st)
    g256 = run('mo_gemm_bf16_256', L.ptr(X), J, J, L.ptr(X), J, 0, L.ptr(dA), N, N, N, J, 0, None, st)
    print(f'J={J:6d} prod: 128-tile {fl/t128/1e9:7.1f} TF ({t128*1e3:.0f} us)  256-ring {fl/t256/1e9:7.1f} TF ({t256*1e3:.0f} us) | '
          f'grad: 128 {fl/g128/1e9:7.1f} TF  256 {fl/g256/1e9:7.1f} TF | bf16-only out: beta0 {fl/tb0/1e9:7.1f} TF ({tb0*1e3:.0f} us) beta1 {fl/tb1/1e9:7.1f} TF')
