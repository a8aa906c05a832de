"""Experiment: how much of the Modified_UNET step is launch / dispatch overhead?  Captures forward + loss + backward of
the config-3 step into one HIP graph (dropout off, optimizer outside) and compares replay with eager launches."""
import os, sys, time, json
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
m.encoder.dropout1.p = 0.0; m.decoder.dropout1.p = 0.0; m.st_gnn.dropout = 0.0
tr = FlatTrainer(m).attach()
x = torch.randn(1, 67, 2, 13, 256, 256, device='cuda'); y = torch.randn_like(x); td = torch.randn(1, 67, 2, 64, device='cuda')

def fb():
    tr.zero_grad()
    out = m(x, td)
    loss, _, _, _ = mse_and_metrics(out, y)
    loss.backward()
    return loss

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

eager = timeit(fb)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): fb()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        loss = fb()
    rep = timeit(g.replay)
    print(json.dumps({"eager_fwd_bwd_ms": round(eager, 2), "graph_replay_ms": round(rep, 2), "loss": float(loss)}))
except Exception as e:
    print('capture failed:', repr(e)[:500]); print(json.dumps({"eager_fwd_bwd_ms": round(eager, 2)}))
