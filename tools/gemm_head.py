"""The head's three bf16 ring-GEMM launches of the throughput mode at the benchmark size (P_f = 3000*256*1 rows):
output-bound shapes (K = 256 / 512) whose fp32-result epilogue decides their time."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_outage_amd._lib as L
L.load()
P, Cs, Ce = 3000 * 256, 256, 512
st = L.stream()
skip_bf = torch.randn(P, Cs, device='cuda').to(torch.bfloat16)
W1_bf = torch.randn(Ce, Cs, device='cuda').to(torch.bfloat16)
b1 = torch.randn(Ce, device='cuda')
r1 = torch.empty(P, Ce, device='cuda')
da1_bf = torch.randn(P, Ce, device='cuda').to(torch.bfloat16)
skip = torch.randn(P, Cs, device='cuda')
dskip = torch.empty(P, Cs, device='cuda')
dskip_bf = torch.empty(P, Cs, device='cuda', dtype=torch.bfloat16)
Wcat_bf = torch.randn(Cs, 256, device='cuda').to(torch.bfloat16)
dg_skip = torch.empty(P, 256, device='cuda')
cases = {
    'end_conv_1 fwd (bias+relu, fp32 out)': lambda: L.call('mo_gemm_bf16_256_ex', L.ptr(skip_bf), Cs, Cs, L.ptr(W1_bf), Cs, 0, L.ptr(r1), Ce, P, Ce, Cs, 0, None, L.ptr(b1), 1, None, st),
    'end_conv_1 dgrad (relu gate, fp32 + bf16 out)': lambda: L.call('mo_gemm_bf16_256_ex', L.ptr(da1_bf), Ce, Ce, L.ptr(W1_bf), Cs, 1, L.ptr(dskip), Cs, P, Cs, Ce, 0, L.ptr(dskip_bf), None, 0, L.ptr(skip), st),
    'skip dgrad all layers (fp32 out)': lambda: L.call('mo_gemm_bf16_256', L.ptr(dskip_bf), Cs, Cs, L.ptr(Wcat_bf), 256, 1, L.ptr(dg_skip), 256, P, 256, Cs, 0, None, st),
}
cases['skip dgrad all layers, 128x128 kernel'] = lambda: L.call('mo_gemm_bf16', L.ptr(dskip_bf), Cs, L.ptr(Wcat_bf), 256, 1, L.ptr(dg_skip), 256, P, 256, Cs, 0, None, st)
bytes_ = {'skip dgrad all layers, 128x128 kernel': P * (Cs * 2 + 256 * 4), 'end_conv_1 fwd (bias+relu, fp32 out)': P * (Cs * 2 + Ce * 4), 'end_conv_1 dgrad (relu gate, fp32 + bf16 out)': P * (Ce * 2 + Cs * 4 + Cs * 6),
          'skip dgrad all layers (fp32 out)': P * (Cs * 2 + 256 * 4)}
for name, fn in cases.items():
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print(f'{name}: {t*1e3:.0f} us, {bytes_[name]/t/1e9:.2f} TB/s of compulsory bytes')
