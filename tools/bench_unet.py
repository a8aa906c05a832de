"""Secondary measurement: Modified_UNET training tiles/s (fwd + MSE + bwd + Adam) on the HIP path.
  python tools/bench_unet.py [--batch 2 --horizon 7 --size 128 --cin 1 --steps 5]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--horizon', type=int, default=7)
    ap.add_argument('--size', type=int, default=128)
    ap.add_argument('--cin', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--opt', action='append', default=[], help='name=value for mo_unet_set_option (A/B switches)')
    ap.add_argument('--dtype', choices=['f32', 'bf16'], default='bf16', help="activation storage (Modified_UNET.act_dtype)")
    ap.add_argument('--main-priority', type=int, default=0, help='run the step on a stream of this priority (-1 = high): the '
                    'weight-gradient lane (priority 0) then only gets what the data-flow chain leaves free (A/B)')
    ap.add_argument('--two-step', action='store_true', help='model(x) -> yhat -> loss kernel instead of the fused tail (A/B)')
    a = ap.parse_args()
    if a.main_priority:
        print('stream priority range', torch.cuda.Stream.priority_range(), file=sys.stderr)
        torch.cuda.set_stream(torch.cuda.Stream(priority=a.main_priority))
    import multimodal_outage_amd._lib as L
    L.load()
    for o in a.opt:
        k, v = o.split('=')
        L.call('mo_unet_set_option', k.encode(), int(v))
    from multimodal_outage_amd.lit import LitModified_UNET
    from multimodal_outage_amd.trainer import FlatTrainer
    torch.manual_seed(42)
    lit = LitModified_UNET('gwnet', a.horizon, 'cuda', input_channels=a.cin, output_channels=a.cin, image_dimension=a.size)
    lit.fused_loss = not a.two_step
    m = lit.model.train()
    m.act_dtype = a.dtype
    tr = FlatTrainer(m, eager_adam=True).attach()      # Adam of a module as soon as its gradients are final (trainer.py)
    B, H, S = a.batch, a.horizon, a.size
    x = torch.randn(B, H, 67, a.cin, S, S, device='cuda')        # the DataLoader's layout (utils.py:101-105)
    y = torch.randn(B, H, 67, a.cin, S, S, device='cuda')
    td = torch.randn(B, 67, H, 64, device='cuda')

    def step():
        tr.zero_grad()
        loss = lit.training_step((x, y, td))                     # lit.py:29-43
        loss.backward()
        tr.allreduce()
        tr.step()
        return loss.detach()       # (the graph behind `loss` holds the step's activations as Function attributes: do not keep it)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs, host = [], 0.0
    for i in range(a.steps):
        h0 = time.perf_counter()
        loss = step()
        host += time.perf_counter() - h0            # time the launch thread needs for a step
        evs.append(torch.cuda.Event()); evs[-1].record()
        if i >= 2:
            evs[i - 2].synchronize()                # at most two steps of lead (keeps the allocator's pool stable)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tiles = B * 67 * H * a.steps
    print(json.dumps({"metric": "UNet (Modified_UNET) train tiles/sec", "value": round(tiles / dt, 1), "unit": "tiles/s",
                      "ms_per_step": round(dt / a.steps * 1e3, 2), "host_launch_ms_per_step": round(host / a.steps * 1e3, 2),
                      "tiles_per_step": B * 67 * H,
                      "config": {"batch": B, "horizon": H, "tile": f"{a.cin}x{S}x{S}", "counties": 67},
                      "loss": round(float(loss), 5), "dtype": a.dtype}))


if __name__ == '__main__':
    main()
