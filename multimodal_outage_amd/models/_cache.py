"""Cached walk of a module tree (named parameters / BatchNorm buffers) that notices when it has gone stale.

The engines need {name: Parameter} and the BatchNorm buffer tensors on every forward; walking the tree costs ~1 ms of
host time per step, so the walk is cached.  nn.Module._apply (``.to()``, ``.cuda()``) on the root drops the cache, but
other operations replace tensors without passing through the root: ``load_state_dict(assign=True)``, ``child.cuda()``
on a sub-module alone (torch's _apply REPLACES buffer objects), assigning a new sub-module, ``register_buffer`` /
``register_parameter``.  Writing BatchNorm running statistics into an orphaned tensor, or reading a stale weight, would
be silent -- so every forward re-checks object identity along the cached tree (a few hundred ``is`` tests, ~50 us) and
re-walks on any mismatch."""
import torch.nn as nn


class TreeCache:
    def __init__(self, root, skip=()):
        """skip: names of top-level children that are not part of this cache (they keep their own)."""
        self.mods = []      # (parent, child name, child)
        self.par = []       # (owner, local name, Parameter)
        self.buf = []       # (owner, local name, buffer tensor)
        self.named = {}
        self.bn = {}
        stack = [('', root)]
        while stack:
            prefix, m = stack.pop()
            for n, p in m._parameters.items():
                self.par.append((m, n, p))
                if p is not None:
                    self.named[prefix + n] = p
            for n, b in m._buffers.items():
                self.buf.append((m, n, b))
            if isinstance(m, nn.BatchNorm2d):
                self.bn[prefix[:-1]] = (m.running_mean, m.running_var, m.num_batches_tracked)
            for n, c in m._modules.items():
                if m is root and n in skip:
                    continue
                self.mods.append((m, n, c))
                if c is not None:
                    stack.append((prefix + n + '.', c))
        self.counts = [(m, len(m._parameters), len(m._buffers), len(m._modules)) for m in
                       [root] + [c for _, _, c in self.mods if c is not None]]

    def valid(self):
        for m, n, c in self.mods:
            if m._modules.get(n) is not c:
                return False
        for m, n, p in self.par:
            if m._parameters.get(n) is not p:
                return False
        for m, n, b in self.buf:
            if m._buffers.get(n) is not b:
                return False
        for m, a, b, c in self.counts:
            if len(m._parameters) != a or len(m._buffers) != b or len(m._modules) != c:
                return False
        return True


def tree_cache(root, attr, skip=()):
    c = root.__dict__.get(attr)
    if c is None or not c.valid():
        c = root.__dict__[attr] = TreeCache(root, skip)
    return c
