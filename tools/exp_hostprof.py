"""Host-side profile of the Modified_UNET step (cProfile over 20 steps after warm-up): where the ~8.7 ms of Python /
ctypes time per step go."""
import cProfile, pstats, os, sys, io
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
def step():
    tr.zero_grad(); out = m(x, td); loss, _, _, _ = mse_and_metrics(out, y); loss.backward(); tr.allreduce(); tr.step()
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
