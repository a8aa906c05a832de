"""Phase timestamps of the small-graph forward kernel (build with -DSG_TIMING: thread 0 of every call writes wall_clock64
deltas -- 100 MHz -- into the unused 6th statistics row).  python tools/sg_timing.py"""
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
