"""Phase timestamps of the small-graph kernels (build gwnet_small.hip with -DSG_TIMING: thread 0 of every call writes
wall_clock64 deltas -- 100 MHz -- into the unused 6th statistics row: forward phases in entries 0..14, the backward's
barriers in 16..).  The timing build is a scratch copy of the library, e.g. on the GPU box:
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -DSG_TIMING -I include -c multimodal_outage_amd/csrc/gwnet_small.hip -o /tmp/gs.o
  (link /tmp/gs.o with the other objects into multimodal_outage_amd/libmo_hip.so of the box's copy);  python tools/sg_timing.py
Round 3, 67 nodes x 2 steps: a backward layer is 72 us = mlp data + weight tiles 21, second hop + dA tiles 22, first hop 7,
gates 6, du + conv weight tiles 8, BatchNorm sums 5, weights 3.5."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_outage_amd.models.graph_wavenet import gwnet
from multimodal_outage_amd import gwnet_engine as E
m = gwnet('cpu', num_nodes=67, dropout=0.0, in_dim=320, out_dim=256, kernel_size=1, horizon=2).cuda().train()
x = torch.randn(1, 67, 2, 320, device='cuda')
grab = {}
real = E.L.call
def spy(name, *a):
    r = real(name, *a)
    if name == 'mo_gwnet_small_fwd':
        grab['stats_ptr'] = a[12]
    return r
E.L.call = spy
for _ in range(3):
    y = m.forward_calls(x)
torch.cuda.synchronize()
# find the stats tensor through ctx
fn = y.grad_fn
while fn is not None and not hasattr(fn, 'keep'):
    fn = fn.next_functions[0][0] if fn.next_functions else None
st = fn.keep[8] if fn is not None else None
print('stats', None if st is None else st.shape)
if st is not None:
    t = st[0, :, 5, :8].cpu() * 10.0      # ns
    names = ['weights', 'A tcn', 'B hop0', 'B hop1', 'C mlp', 'stats']
    raw = st[0, :, 5, :].cpu()
    print('shader clock during the kernel (MHz):', [round(float(raw[li, 8] / raw[li, 9] * 100.0)) for li in range(raw.shape[0])])
    print('phase A of wave 0, cycles: loads %d, mfma %d, epilogue %d, barrier wait %d, start offset %d' % tuple(int(raw[3, k]) for k in (10, 11, 12, 13, 14)))
    for li in range(t.shape[0]):
        v = t[li].tolist()
        d = [v[0]] + [v[i] - v[i - 1] for i in range(1, 6)]
        print('layer', li, ' '.join(f'{n}={x_ / 1e3:.2f}us' for n, x_ in zip(names, d)))
# backward: stamps after every barrier of a layer (sg_bwd_kernel, same build flag), entries 16.. of the same row
y.float().sum().backward()
torch.cuda.synchronize()
if st is not None:
    rawb = st[0, :, 5, 16:28].cpu() * 10.0
    full = ['weights', 'R1 bn sums', 'R1 fold', 'R2 dh', 'M mlp + dWm', 'N1 hop', 'N2 hop + dA', 'T1 gates', 'T2 du + dW']
    last = ['weights', 'skip only', 'T1 gates', 'T2 du + dW']
    tot = 0.0
    for li in range(rawb.shape[0] - 1, -1, -1):
        names = last if li == rawb.shape[0] - 1 else full
        v = rawb[li, :len(names)].tolist()
        d = [v[0]] + [v[i] - v[i - 1] for i in range(1, len(names))]
        tot += v[len(names) - 1]
        print('bwd layer', li, ' '.join(f'{n}={x_ / 1e3:.2f}' for n, x_ in zip(names, d)), f'| {v[len(names) - 1] / 1e3:.1f} us')
    print(f'bwd layers total {tot / 1e3:.1f} us')
if st is not None:
    sub = st[0, :, 5, 16:32].cpu() * 10.0
    for li in (3, 2):
        print(f'bwd layer {li}: M strips done at +{(sub[li, 12] - sub[li, 3]) / 1e3:.2f} us of the phase, N2 hop jobs done at +{(sub[li, 13] - sub[li, 5]) / 1e3:.2f} us (wave 0)')
