"""world_size-2 gloo test of the data-parallel plumbing (runs on CPU): rank-0 broadcast of the flat
parameter buffer and buffers, gradient all-reduce (explicit and bucketed from inside backward) with 1/world
averaging, identical handling of never-touched (zero) gradients on all ranks."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from multimodal_outage_amd.trainer import FlatTrainer
        torch.manual_seed(100 + rank)                 # ranks start with DIFFERENT weights
        m = nn.Sequential(nn.Conv2d(3, 5, 1), nn.BatchNorm2d(5), nn.Linear(7, 2))
        m[1].running_mean.fill_(float(rank))
        tr = FlatTrainer(m)
        # after construction every rank holds rank 0's parameters and buffers
        ref = [torch.zeros_like(tr.flat_p) for _ in range(world)]
        dist.all_gather(ref, tr.flat_p)
        same = all(torch.equal(ref[0], r) for r in ref)
        buf_ok = float(m[1].running_mean[0]) == 0.0
        # parameters are views of the flat buffer; grads are views of the flat grad buffer
        views_ok = all(p.data_ptr() >= tr.flat_p.data_ptr() for p in m.parameters())
        # rank-dependent gradients; one parameter (the Linear bias) is never touched -> stays zero
        for k, g in tr.grad_views.items():
            if k != '2.bias':
                g.fill_(float(rank + 1))
        tr.allreduce()
        summed = {k: float(g.flatten()[0]) for k, g in tr.grad_views.items()}
        # overlapped path: a real backward fires the post-accumulate hooks, every top-level child is announced as a
        # bucket from inside backward, allreduce() only waits; the result must be the sum of both ranks' gradients
        import copy
        refm = copy.deepcopy(m)
        for p_ in refm.parameters():
            p_.grad = None
        tr.zero_grad()
        xs = [torch.full((2, 3, 4, 7), float(r + 1)) + torch.arange(7.0) for r in range(world)]
        want = {k: torch.zeros_like(v) for k, v in refm.named_parameters()}
        for r in range(world):
            refm2 = copy.deepcopy(refm)
            refm2(xs[r]).square().sum().backward()
            for k, v in refm2.named_parameters():
                want[k] += v.grad
        bucket_ok = True
        for it in range(3):
            # the first backward pass only learns which parameters fire (0 announcements), later passes announce
            # every top-level child from inside backward
            tr.zero_grad()
            m(xs[rank]).square().sum().backward()
            announced = len(tr._done)
            tr.allreduce()
            bucket_ok = bucket_ok and announced == (0 if it == 0 else 3) and all(
                torch.allclose(tr.grad_views[k], want[k], rtol=1e-5, atol=1e-5) for k in want)
        q.put((rank, same, buf_ok, views_ok, summed, tr.world, bucket_ok))
    finally:
        dist.destroy_process_group()


def test_flat_trainer_gloo_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, buf_ok, views_ok, summed, w, bucket_ok in res:
        assert same and buf_ok and views_ok and w == 2 and bucket_ok
        for k, v in summed.items():
            assert v == (0.0 if k == '2.bias' else 3.0), (k, v)     # 1 + 2 summed; Adam divides by world


class _Branchy(nn.Module):
    """A bucket ('body') holding parameters that never enter the autograd graph (as gwnet's residual_convs.* with gcn
    on inside Modified_UNET's 'st_gnn' bucket), one used twice, and a second bucket ('tail')."""

    def __init__(self):
        super().__init__()
        self.body = nn.ModuleDict(dict(a=nn.Linear(6, 6), unused=nn.Linear(6, 6), b=nn.Linear(6, 6)))
        self.tail = nn.Linear(6, 3)

    def forward(self, x):
        h = torch.tanh(self.body['a'](x))
        h = self.body['b'](h) + self.body['a'](h)
        return self.tail(h)


def _worker_steps(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from multimodal_outage_amd.trainer import FlatTrainer
        torch.manual_seed(5)
        m = _Branchy()
        tr = FlatTrainer(m)
        calls = []
        orig = tr.mark_ready
        tr.mark_ready = lambda names: (calls.append(tuple(names)), orig(names))[1]
        # gradients must be complete when a bucket is announced: snapshot flat_g at every announcement and compare
        # the announced ranges with the end-of-backward local gradient
        worst, counts, premature = 0.0, [], 0
        for it in range(5):
            tr.zero_grad()
            calls.clear()
            snaps = []
            tr.mark_ready = lambda names: (snaps.append((tuple(names), tr.flat_g.clone())), orig(names))[1]
            x = torch.randn(4, 6, generator=torch.Generator().manual_seed(1000 * it + rank))
            m(x).square().sum().backward()
            # local gradient of this rank, recomputed without the trainer
            import copy
            ref = copy.deepcopy(m)
            for p_ in ref.parameters():
                p_.grad = None
            ref(x).square().sum().backward()
            local = torch.zeros_like(tr.flat_g)
            for k, p_ in ref.named_parameters():
                lo, _ = tr._span[k]
                if p_.grad is not None:
                    local[lo:lo + p_.numel()] = p_.grad.reshape(-1)
            for names, snap in snaps:
                for k in names:
                    lo, hi = tr._span[k]
                    # gloo reduces in place asynchronously, so compare the snapshot taken BEFORE the collective
                    if not torch.equal(snap[lo:hi], local[lo:hi]):
                        premature += 1
            counts.append(len(snaps))
            tr.allreduce()
            parts = [torch.zeros_like(local) for _ in range(world)]
            dist.all_gather(parts, local)
            want = sum(parts)
            worst = max(worst, float((tr.flat_g - want).abs().max()))
            tr.step_count = 0                       # no optimizer step: mo_adam_step is a HIP kernel
        q.put((rank, worst, counts, premature))
    finally:
        dist.destroy_process_group()


def test_flat_trainer_buckets_rearm_with_never_fired_parameters():
    """a bucket containing parameters that never fire must not drift across steps -- 5 steps,
    flat_g == all-gathered sum of local gradients at every step, no announcement before a bucket is complete."""
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_steps, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, worst, counts, premature in res:
        assert premature == 0, (rank, premature)
        assert worst <= 1e-6, (rank, worst)
        assert counts == [0, 2, 2, 2, 2], counts      # learning pass, then both buckets from inside backward


def test_bench_launches_its_own_ranks_when_started_plainly():
    """`python bench.py --gpus 2` (no torchrun, WORLD_SIZE unset) must not die on an assert: it becomes the
    launcher -- child process `python -m torch.distributed.run --nproc-per-node 2 bench.py ...` -- and passes rank 0's line
    through.  --rehearse keeps the ranks off the GPU (rendezvous, barrier, max-over-ranks timing, one JSON line), so the
    launcher's control flow runs here on the CPU with gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--backend', 'gloo', '--rehearse'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout                      # ONE line, from rank 0
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['steps'] == 3 and line['rehearsal'] is True
    # max over ranks: rank 1 sleeps 2 ms per step, rank 0 only 1 ms
    assert line['ms_per_step'] >= 2.0
    # a failing rank's exit code comes back through the launcher
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--backend', 'no_such_backend',
                        '--rehearse'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


def test_flat_buffer_spans_split_evenly_for_any_world_size(monkeypatch):
    """reduce-scatter + all-gather needs every reduced range to be a multiple of the world size: spans are aligned to
    lcm(64, world) floats (round 2 required n % world == 0 and nothing padded for it)."""
    from multimodal_outage_amd import trainer as T
    m = nn.Sequential(nn.Conv2d(3, 5, 1), nn.BatchNorm2d(5), nn.Linear(7, 3))
    for world in (1, 2, 3, 6, 8):
        monkeypatch.setattr(T.dist, 'is_initialized', lambda: False)
        tr = T.FlatTrainer.__new__(T.FlatTrainer)
        # (only the layout arithmetic: a real group of `world` ranks is not needed for it)
        al = T.math.lcm(T._ALIGN, world)
        total = 0
        for p in m.parameters():
            total += (p.numel() + al - 1) // al * al
        assert total % world == 0 and al % world == 0 and al % 64 == 0
