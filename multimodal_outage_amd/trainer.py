"""Data-parallel training plumbing for the HIP path: flat fp32 parameter / gradient / Adam-state
buffers, one-bucket RCCL all-reduce, fused Adam (torch.optim.Adam semantics, lit.py:59-61).

One process per GPU; ``torch.distributed`` backend "nccl" is RCCL on ROCm (xGMI inside a node).
The reference has no explicit distributed code (Lightning's default DDP, lit.py:204): this is the
MI355X-native equivalent -- replicas with per-rank BatchNorm statistics (no SyncBN, as in the
reference), gradients averaged across ranks each step.  Parameters that never receive a gradient
(residual_convs.*, gconv.7, bn.7 -- SURVEY.md 3.4) keep zero gradients, identically on all ranks.
"""
import math

import torch
import torch.distributed as dist

from . import _lib as L

_ALIGN = 64   # floats (256 B): every parameter starts on a 16-byte boundary for vector loads


class FlatTrainer:
    def __init__(self, module, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, process_group=None):
        self.module = module
        self.lr, self.betas, self.eps = lr, betas, eps
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        params = [(k, p) for k, p in module.named_parameters()]
        dev = params[0][1].device
        offs, total = [], 0
        for _, p in params:
            offs.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.flat_p = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(total, device=dev, dtype=torch.float32)
        self.m = torch.zeros(total, device=dev, dtype=torch.float32)
        self.v = torch.zeros(total, device=dev, dtype=torch.float32)
        self.grad_views = {}
        for (k, p), o in zip(params, offs):
            n = p.numel()
            self.flat_p[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[o:o + n].view(p.shape)
            gv = self.flat_g[o:o + n].view(p.shape)
            p.grad = gv                      # autograd accumulates in place for anything not direct-written
            self.grad_views[k] = gv
        self.step_count = 0
        self.total = total
        if self.world > 1:
            # replicas start identical: rank-0 broadcast of parameters and buffers (DDP default)
            dist.broadcast(self.flat_p, 0, group=self.pg)
            for b in module.buffers():
                dist.broadcast(b, 0, group=self.pg)

    def grad_out(self, prefix=''):
        """{name-without-prefix: grad view} for an engine that writes gradients in place."""
        return {k[len(prefix):]: v for k, v in self.grad_views.items() if k.startswith(prefix)}

    def zero_grad(self):
        self.flat_g.zero_()

    def allreduce(self, async_op=False):
        if self.world > 1:
            return dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=self.pg, async_op=async_op)
        return None

    def step(self):
        """Adam on the flat buffer; gradients are averaged over ranks (grad_scale = 1/world)."""
        self.step_count += 1
        b1, b2 = self.betas
        bc1 = 1.0 - b1 ** self.step_count
        bc2 = 1.0 - b2 ** self.step_count
        L.call('mo_adam_step', L.ptr(self.flat_p), L.ptr(self.flat_g), L.ptr(self.m), L.ptr(self.v),
               self.total, self.lr, b1, b2, self.eps, bc1, bc2, 1.0 / self.world, L.stream())

    def set_lr(self, lr):
        self.lr = lr


def cosine_lr(base_lr, epoch, t_max=10, eta_min=0.0):
    """CosineAnnealingLR(T_max=10) closed form (lit.py:61), stepped per epoch."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch / t_max)) / 2
