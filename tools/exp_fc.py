import sys, os, torch, ctypes
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import multimodal_outage_amd._lib as L
lib = L.load()
dev = 'cuda'
P = 134
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
for (Co, Ci) in ((4096, 16384), (16384, 1024)):
    W = torch.randn(Co, Ci, device=dev) * 0.01; b = torch.randn(Co, device=dev)
    x = torch.randn(P, Ci, device=dev); dout = torch.randn(P, Co, device=dev)
    out = torch.empty(P, Co, device=dev); din = torch.empty(P, Ci, device=dev)
    ws = torch.empty(max(lib.mo_linear_splitk_ws_floats(P, Co, Ci), lib.mo_linear_splitk_ws_floats(P, Ci, Co)), device=dev)
    f = t(lambda: L.call('mo_conv1x1_fwd_splitk', L.ptr(x), Ci, L.ptr(W), L.ptr(b), Co, L.ptr(out), P, 1, L.ptr(ws), L.stream()))
    d = t(lambda: L.call('mo_conv1x1_bwd_data_splitk', L.ptr(dout), Co, P, L.ptr(W), Ci, L.ptr(din), L.ptr(ws), L.stream()))
    ws3 = torch.empty(max(lib.mo_fc3_ws_floats(P, Ci, Co), lib.mo_fc3_ws_floats(P, Co, Ci)), device=dev)
    f3 = t(lambda: L.call('mo_fc3_fwd', L.ptr(x), P, Ci, L.ptr(W), L.ptr(b), Co, 1, L.ptr(out), L.ptr(ws3), L.stream()))
    d3 = t(lambda: L.call('mo_fc3_bwd_data', L.ptr(dout), P, Co, L.ptr(W), Ci, L.ptr(din), L.ptr(ws3), L.stream()))
    dW = torch.empty(Co, Ci, device=dev); db = torch.empty(Co, device=dev)
    wsw = torch.empty(lib.mo_wgrad_ws_floats(Co, Ci, P), device=dev)
    w = t(lambda: L.call('mo_conv1x1_bwd_weight', L.ptr(dout), Co, P, L.ptr(x), Ci, 0, 0, 0, 0, L.ptr(dW), L.ptr(db), L.ptr(wsw), L.stream()))
    wsw3 = torch.empty(lib.mo_fc3_wgrad_ws_floats(P, Co, Ci), device=dev)
    w3 = t(lambda: L.call('mo_fc3_bwd_weight', L.ptr(dout), P, Co, L.ptr(x), Ci, L.ptr(dW), L.ptr(db), L.ptr(wsw3), L.stream()))
    print(f'   weight gradient: exact {w:.1f} us, 3 x bf16 {w3:.1f} us')
    print(f'W {Co}x{Ci}: exact fp32 fwd {f:.1f} us, dgrad {d:.1f} us;  3 x bf16 fwd {f3:.1f} us, dgrad {d3:.1f} us', flush=True)
