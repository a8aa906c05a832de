"""Host-side profile of the Modified_UNET step: (a) launch-thread time per phase (forward / loss / backward / optimizer,
no synchronisation inside), (b) cProfile of forward and -- through threading.setprofile -- of the autograd thread that runs
the Functions' backward."""
import cProfile, pstats, os, sys, io, time, threading
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_outage_amd._lib as L
L.load()
from multimodal_outage_amd.models.unet import Modified_UNET
from multimodal_outage_amd.lit import mse_and_metrics
from multimodal_outage_amd.trainer import FlatTrainer
torch.manual_seed(42)
m = Modified_UNET('gwnet', 2, input_channels=13, output_channels=13, image_dimension=256).cuda().train()
m.act_dtype = 'bf16'
tr = FlatTrainer(m).attach()
x = torch.randn(1, 67, 2, 13, 256, 256, device='cuda'); y = torch.randn_like(x); td = torch.randn(1, 67, 2, 64, device='cuda')
T = {'zero': 0.0, 'fwd': 0.0, 'loss': 0.0, 'bwd': 0.0, 'opt': 0.0}
def step(acc=None):
    t0 = time.perf_counter(); tr.zero_grad()
    t1 = time.perf_counter(); out = m(x, td)
    t2 = time.perf_counter(); loss, _, _, _ = mse_and_metrics(out, y)
    t3 = time.perf_counter(); loss.backward()
    t4 = time.perf_counter(); tr.allreduce(); tr.step()
    t5 = time.perf_counter()
    if acc is not None:
        for k, v in zip(('zero', 'fwd', 'loss', 'bwd', 'opt'), (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): acc[k] += v
for _ in range(3): step()
torch.cuda.synchronize()
N = 20
for _ in range(N): step(T)
torch.cuda.synchronize()
print({k: round(v / N * 1e3, 2) for k, v in T.items()}, 'ms per step (host)')
calls = {}
orig = L.call
def spy(nm, *a):
    calls[nm] = calls.get(nm, 0) + 1
    return orig(nm, *a)
L.call = spy
import multimodal_outage_amd.unet_engine as ue, multimodal_outage_amd.gwnet_engine as ge
step(); torch.cuda.synchronize(); L.call = orig
print('C-ABI launches per step:', sum(calls.values()), sorted(calls.items(), key=lambda kv: -kv[1])[:12])
# (b) cProfile of the two Functions' backward (they run on the autograd engine's thread: the profiler is switched on inside them)
pr = cProfile.Profile()
for Fn in (ue.UnetDecodeFn, ue.UnetEncodeFn):
    ob = Fn.backward
    def wrap(ctx, *g, _o=ob):
        pr.enable()
        try:
            return _o(ctx, *g)
        finally:
            pr.disable()
    Fn.backward = staticmethod(wrap)
for _ in range(10):
    step()
torch.cuda.synchronize()
sio = io.StringIO()
pstats.Stats(pr, stream=sio).sort_stats('tottime').print_stats(28)
print(sio.getvalue()[:6000])
pf = cProfile.Profile()
pf.enable()
for _ in range(10):
    out = m(x, td)
pf.disable()
torch.cuda.synchronize()
sio = io.StringIO()
pstats.Stats(pf, stream=sio).sort_stats('tottime').print_stats(18)
print(sio.getvalue()[:4000])
