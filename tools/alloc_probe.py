import sys, torch, gc
sys.path.insert(0, '/root/repo')
from multimodal_outage_amd.lit import LitModified_UNET
from multimodal_outage_amd.trainer import FlatTrainer
torch.manual_seed(42)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lit = LitModified_UNET('gwnet', 2, 'cuda', input_channels=13, output_channels=13, image_dimension=256)
m = lit.model.train(); m.act_dtype = 'bf16'
tr = FlatTrainer(m, eager_adam=(len(sys.argv) > 3)).attach()
x = torch.randn(B, 2, 67, 13, 256, 256, device='cuda'); y = torch.randn_like(x); td = torch.randn(B, 67, 2, 64, device='cuda')
prev = 0
lead = int(sys.argv[2]) if len(sys.argv) > 2 else 0        # 0: synchronise every step; k: at most k steps of lead
evs = []
for i in range(40):
    tr.zero_grad(); loss = lit.training_step((x, y, td)); loss.backward(); tr.allreduce(); tr.step()
    evs.append(torch.cuda.Event()); evs[-1].record()
    if lead == 0:
        torch.cuda.synchronize()
    elif i >= lead:
        evs[i - lead].synchronize()
    s = torch.cuda.memory_stats()
    n = s.get('num_device_alloc', 0)
    print(i, 'allocs', n - prev, 'reserved MB', s['reserved_bytes.all.current'] >> 20, 'active MB', s['active_bytes.all.peak'] >> 20, 'gc', gc.get_count())
    prev = n
