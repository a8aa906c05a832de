"""Data-parallel training plumbing for the HIP path: flat fp32 parameter / gradient / Adam-state
buffers, bucketed RCCL all-reduce overlapped with backward, fused Adam (torch.optim.Adam semantics, lit.py:59-61).

One process per GPU; ``torch.distributed`` backend "nccl" is RCCL on ROCm (xGMI inside a node).
The reference has no explicit distributed code (Lightning's default DDP, lit.py:204): this is the
MI355X-native equivalent -- replicas with per-rank BatchNorm statistics (no SyncBN, as in the
reference), gradients averaged across ranks each step.  Parameters that never receive a gradient
(residual_convs.*, gconv.7, bn.7 -- SURVEY.md 3.4) keep zero gradients, identically on all ranks.

Overlap: the engines call ``mark_ready(names)`` from inside backward as soon as a group of gradients is final
(end of each autograd Function; Graph-WaveNet additionally after the late half of its layers).  The covering
ranges of the flat gradient buffer are all-reduced asynchronously right there -- RCCL orders the collective behind
the work already queued on the calling stream and runs it beside the rest of backward -- and ``allreduce()`` at
the end of the step reduces whatever was not announced and waits for the handles.  One backward pass per
allreduce(): a range whose all-reduce has started must not be accumulated into again before the wait.
"""
import math

import torch
import torch.distributed as dist

from . import _lib as L

_ALIGN = 64   # floats (256 B): every parameter starts on a 16-byte boundary for vector loads


def _join_deferred_lanes():
    from . import unet_engine
    if unet_engine._Lane._deferred:
        unet_engine.join_deferred()


class FlatTrainer:
    def __init__(self, module, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, process_group=None, overlap=True,
                 force_collectives=False, collective='allreduce', comm='torch', eager_adam=False):
        self.module = module
        self.overlap = overlap   # False: nothing starts inside backward, allreduce() reduces the whole buffer
        self.lr, self.betas, self.eps = lr, betas, eps
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # collectives run when there is more than one rank -- or when forced (a world_size-1 RCCL communicator on a
        # one-GPU box exercises librccl, the asynchronous handles and the stream ordering; tests/test_rccl_gpu.py)
        self.collectives = self.world > 1 or (force_collectives and dist.is_initialized())
        # 'allreduce': one ncclAllReduce per bucket.  'rs_ag': reduce-scatter + all-gather of the bucket (SURVEY 8e:
        # on the fully connected xGMI mesh each rank then exchanges its 1/world shard with every peer over its own
        # link instead of walking a ring); RCCL backend only (gloo has no reduce-scatter).
        assert collective in ('allreduce', 'rs_ag')
        self.collective = collective
        # comm='torch': torch.distributed collectives on `process_group` (backend "nccl" = RCCL).  comm='abi': the
        # library's own RCCL communicator (include/mo_hip.h mo_allreduce_*: the collective runs on its own HIP stream,
        # ordered by events behind the producing stream and in front of the Adam kernel); torch.distributed is then
        # only the host channel that hands rank 0's unique id to the other ranks.
        assert comm in ('torch', 'abi')
        self.comm = comm
        self._abi = None
        if comm == 'abi' and self.collectives:
            import ctypes as C
            ident = [None]
            if dist.get_rank(process_group) == 0:
                raw = C.create_string_buffer(128)
                L.call('mo_allreduce_unique_id', raw)
                ident = [raw.raw]
            # `src` of broadcast_object_list is a GLOBAL rank: group rank 0 of a sub-group need not be global rank 0
            src = dist.get_global_rank(process_group, 0) if process_group is not None else 0
            dist.broadcast_object_list(ident, src=src, group=process_group)
            h = C.c_void_p()
            L.call('mo_allreduce_init', C.create_string_buffer(ident[0], 128), dist.get_rank(process_group),
                   self.world, C.byref(h))
            self._abi = h
        params = [(k, p) for k, p in module.named_parameters()]
        dev = params[0][1].device
        # every parameter's span is a multiple of lcm(64 floats, world): any run of adjacent parameters (a bucket, or the
        # whole buffer) then splits evenly over the ranks, which the reduce-scatter + all-gather form needs -- for every
        # world size, not only the powers of two that divide the 64-float alignment.  Padding holds zeros and stays zero.
        al = self._al = math.lcm(_ALIGN, max(self.world, 1))
        offs, total = [], 0
        for _, p in params:
            offs.append(total)
            total += (p.numel() + al - 1) // al * al
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
        # eager Adam (single process only): an engine announces from inside backward that the gradients of whole top-level
        # modules are final (adam_now); their Adam update is launched right there, on the engine's weight-gradient lane,
        # beside the rest of backward -- the 85 M parameters of Modified_UNET's FC bottleneck (95 % of Adam's 2.4 GB of
        # traffic) are final 1.5 ms before the step ends.  step() then updates what is left.  With collectives the update
        # has to wait for the all-reduce, so it stays at the end of the step.
        # Opt-in: parameters change DURING backward, so every backward pass must be followed by step().
        self.eager_adam = bool(eager_adam) and not self.collectives
        self._stepped = []       # [lo, hi) ranges already updated in this step
        self._overwritten = []   # [lo, hi) ranges an attached engine overwrites in every backward pass (see zero_grad)
        self._span = {k: (o, o + (p.numel() + al - 1) // al * al) for (k, p), o in zip(params, offs)}
        self._done = []          # [lo, hi) ranges already handed to an asynchronous all-reduce this step
        self._work = []
        # buckets = top-level child modules, announced from the post-accumulate hook of the LAST parameter of the
        # bucket that fires in a backward pass.  Which parameters fire is a property of the autograd graph, not of
        # requires_grad (residual_convs.* with gcn on, gconv.7/bn.7 never enter it; this torch fires the hook of a
        # finished Function's AccumulateGrad nodes even for None gradients), so the firing set of every bucket is
        # LEARNED in the first backward pass (no hook-driven announcements there: the final pass of allreduce() covers
        # everything) and every later pass re-arms from it.  Parameters an engine writes directly are announced by the
        # engine itself through ready_callback, earlier than their hooks would.
        self._bucket = {k: k.split('.')[0] for k, _ in params}
        self._bucket_names = {}
        for k, p in params:
            self._bucket_names.setdefault(self._bucket[k], []).append(k)
        self._expected = None        # {bucket: frozenset(names that fired in the previous backward pass)}
        self._fired = {}             # {bucket: set(names fired in this backward pass)}
        self._announced = set()      # buckets handed to mark_ready by the hooks in this backward pass
        if self.collectives and hasattr(torch.Tensor, 'register_post_accumulate_grad_hook'):
            for k, p in params:
                if p.requires_grad:
                    p.register_post_accumulate_grad_hook(lambda _p, k=k: self._on_grad(k))
        if self.world > 1:
            # replicas start identical: rank-0 broadcast of parameters and buffers (DDP default)
            dist.broadcast(self.flat_p, 0, group=self.pg)
            for b in module.buffers():
                dist.broadcast(b, 0, group=self.pg)

    def grad_out(self, prefix=''):
        """{name-without-prefix: grad view} for an engine that writes gradients in place."""
        return {k[len(prefix):]: v for k, v in self.grad_views.items() if k.startswith(prefix)}

    def attach(self):
        """Let the engines write gradients straight into the flat gradient buffer (no autograd accumulation kernels):
        Graph-WaveNet as the top-level module, and the UNet-side Functions of Modified_UNET.  One forward/backward
        per step (an engine overwrites its gradients, it does not accumulate across several backward passes)."""
        from .models.graph_wavenet import gwnet
        from .models.unet import Modified_UNET
        m = self.module
        if isinstance(m, gwnet):
            m._mo_grad_out = self.grad_out()
            m._mo_grad_ready = self.ready_callback()
        elif isinstance(m, Modified_UNET):
            m._mo_grad_out = {k: v for k, v in self.grad_views.items() if not k.startswith('st_gnn.')}
            # the inner Graph WaveNet is called once per batch element (unet.py:221): Modified_UNET.forward hands it these
            # views only for a batch of one window (82 AccumulateGrad adds per step otherwise); with more windows its
            # gradients accumulate through autograd
            m._mo_grad_out_st_gnn = self.grad_out('st_gnn.')
            m._mo_adam_now = self.adam_now if self.eager_adam else None
            # ranges the engine overwrites in every backward pass (zero_grad skips them), merged
            runs = []
            for lo, hi in sorted(v for k, v in self._span.items() if not k.startswith('st_gnn.')):
                if runs and lo <= runs[-1][1]:
                    runs[-1][1] = max(runs[-1][1], hi)
                else:
                    runs.append([lo, hi])
            self._overwritten = [tuple(r) for r in runs]
        return self

    def zero_grad(self, full=False):
        """Zero the flat gradient buffer.  After attach() on a Modified_UNET the UNet-side ranges are skipped unless
        `full`: the engine OVERWRITES every one of those gradients in each backward pass (attach()), so the 330 MB fill of
        config 3 -- 41 us at the head of every step's main stream -- only ever cleared values about to be replaced.  (A
        step() without a backward pass in between then re-applies the previous gradients there instead of zeros.)"""
        if full or not self._overwritten:
            self.flat_g.zero_()
            return
        pos = 0
        for lo, hi in self._overwritten + [(self.total, self.total)]:
            if lo > pos:
                self.flat_g[pos:lo].zero_()
            pos = max(pos, hi)

    def ready_callback(self, prefix=''):
        """Callback for an engine: cb(names) announces that the gradients of `names` (without `prefix`) are final."""
        return lambda names: self.mark_ready([prefix + n for n in names])

    def _on_grad(self, k):
        b = self._bucket[k]
        if b in self._announced:
            # its bucket's all-reduce has started: this gradient would land beside / after the collective
            raise RuntimeError(f'FlatTrainer: gradient of {k!r} arrived after bucket {b!r} was all-reduced (the set of '
                               'parameters receiving gradients changed between steps, or backward ran twice before '
                               'allreduce()); use FlatTrainer(overlap=False) for such a model')
        fired = self._fired.setdefault(b, set())
        fired.add(k)
        if self._expected is None or not self.overlap:
            return                               # first pass: learn the firing sets only
        exp = self._expected.get(b)
        if exp is not None and fired == exp:     # an unexpected name keeps the sets unequal -> final pass, re-learn
            self._announced.add(b)
            self.mark_ready(self._bucket_names[b])

    def _reduce(self, lo, hi):
        """Asynchronous sum over ranks of flat_g[lo:hi]; returns the handles to wait for."""
        buf = self.flat_g[lo:hi]
        rs = not (self.collective == 'allreduce' or (hi - lo) % self.world != 0 or (hi - lo) < 65536)
        if self._abi is not None:
            L.call('mo_allreduce_launch', self._abi, buf.data_ptr(), hi - lo, int(rs), L.stream())
            return []
        if not rs:
            return [dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)]
        n = (hi - lo) // self.world
        rank = dist.get_rank(self.pg)
        shard = buf[rank * n:(rank + 1) * n]          # reduced in place into this rank's slice, then gathered
        w1 = dist.reduce_scatter_tensor(shard, buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        w2 = dist.all_gather_into_tensor(buf, shard, group=self.pg, async_op=True)   # same communicator: ordered
        return [w1, w2]

    def mark_ready(self, names):
        """Start the all-reduce of the flat-gradient ranges covering `names` (maximal runs of adjacent parameters);
        call on the stream the gradients were produced on.  No-op for a single process."""
        if not self.collectives or not self.overlap:
            return
        spans = sorted(self._span[k] for k in names if k in self._span)
        runs = []
        for lo, hi in spans:
            if runs and lo <= runs[-1][1]:
                runs[-1][1] = max(runs[-1][1], hi)
            else:
                runs.append([lo, hi])
        for lo, hi in runs:
            if any(lo < dhi and dlo < hi for dlo, dhi in self._done):
                continue                     # (part of) the run was announced before: leave it to the final pass
            self._work += self._reduce(lo, hi)
            self._done.append((lo, hi))

    def allreduce(self, async_op=False):
        """Reduce every range not announced through mark_ready, then wait for the asynchronous buckets."""
        _join_deferred_lanes()
        if not self.collectives:
            return None
        done = sorted(self._done)
        pos = 0
        for lo, hi in done + [(self.total, self.total)]:
            if lo > pos:
                self._work += self._reduce(pos, lo)
            pos = max(pos, hi)
        for w in self._work:
            w.wait()
        if self._abi is not None:
            L.call('mo_allreduce_wait', self._abi, L.stream())
        self._work, self._done = [], []
        # re-arm every bucket for the next backward pass from what fired in this one
        if self._fired:
            self._expected = {b: frozenset(f) for b, f in self._fired.items()}
        self._fired, self._announced = {}, set()
        return None

    def _adam(self, lo, hi, step_no, stream):
        b1, b2 = self.betas
        bc1 = 1.0 - b1 ** step_no
        bc2 = 1.0 - b2 ** step_no
        off = lo * 4
        L.call('mo_adam_step', self.flat_p.data_ptr() + off, self.flat_g.data_ptr() + off, self.m.data_ptr() + off,
               self.v.data_ptr() + off, hi - lo, self.lr, b1, b2, self.eps, bc1, bc2, 1.0 / self.world, stream)

    def adam_now(self, prefixes, stream=None):
        """Engine callback (inside backward): every gradient of the parameters whose names start with one of `prefixes`
        is final -- update them now on `stream` (default: the current stream).  No-op unless eager_adam."""
        if not self.eager_adam:
            return
        spans = sorted(v for k, v in self._span.items() if any(k.startswith(p_) for p_ in prefixes))
        runs = []
        for lo, hi in spans:
            if runs and lo <= runs[-1][1]:
                runs[-1][1] = max(runs[-1][1], hi)
            else:
                runs.append([lo, hi])
        st = L.stream() if stream is None else stream
        for lo, hi in runs:
            if any(lo < dhi and dlo < hi for dlo, dhi in self._stepped):
                continue
            self._adam(lo, hi, self.step_count + 1, st)
            self._stepped.append((lo, hi))

    def step(self):
        """Adam on the flat buffer (what adam_now has not updated already in this step); gradients are averaged over
        ranks (grad_scale = 1/world)."""
        _join_deferred_lanes()
        self.step_count += 1
        pos = 0
        for lo, hi in sorted(self._stepped) + [(self.total, self.total)]:
            if lo > pos:
                self._adam(pos, lo, self.step_count, L.stream())
            pos = max(pos, hi)
        self._stepped = []

    def set_lr(self, lr):
        self.lr = lr

    def close(self):
        """Release the library's RCCL communicator, its stream and events (comm='abi').  Also run by __del__ and on
        leaving a `with FlatTrainer(...) as tr:` block."""
        if self._abi is not None:
            h, self._abi = self._abi, None
            L.call('mo_allreduce_destroy', h)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:          # interpreter shutdown: the library may already be gone
            pass


def cosine_lr(base_lr, epoch, t_max=10, eta_min=0.0):
    """CosineAnnealingLR(T_max=10) closed form (lit.py:61), stepped per epoch."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch / t_max)) / 2
