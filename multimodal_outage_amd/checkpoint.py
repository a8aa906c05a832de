"""Checkpoint compatibility with the reference's Lightning run.

The reference trains under ``L.Trainer`` with ``ModelCheckpoint`` (lit.py:187-196): a ``.ckpt`` is a dict with
``state_dict`` (keys ``model.<name>`` -- LitModified_UNET holds the network as ``self.model``, lit.py:23),
``optimizer_states[0]`` (``torch.optim.Adam.state_dict()``: per-parameter ``exp_avg`` / ``exp_avg_sq`` / ``step``
indexed in ``parameters()`` order, lit.py:60), ``lr_schedulers[0]`` (``CosineAnnealingLR.state_dict()``, lit.py:61) and
``epoch`` / ``global_step``.  The product keeps the reference's state_dict keys and shapes, so weights load with
``load_state_dict``; this module adds the rest: prefix handling, and the optimizer / scheduler state mapped into
``FlatTrainer``'s flat Adam buffers so that a resumed step continues the reference run.

Files are read with ``torch.load(..., weights_only=True)`` only (nothing from the file is executed).
"""
import torch

from .trainer import cosine_lr


def read_checkpoint(path, map_location='cpu'):
    """Load a Lightning .ckpt (or a bare state_dict file) without executing anything from it."""
    return torch.load(path, map_location=map_location, weights_only=True)


def _strip(sd, prefix):
    if prefix and any(k.startswith(prefix) for k in sd):
        return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    return dict(sd)


def load_lightning_state(ckpt, module, trainer=None, prefix='model.', strict=True, base_lr=None, t_max=10):
    """Load a reference checkpoint into ``module`` (and its optimizer / scheduler state into ``trainer``).

    ckpt     a Lightning checkpoint dict (``state_dict`` [+ ``optimizer_states``, ``lr_schedulers``]) or a bare
             state_dict; a path is read with ``read_checkpoint``.
    module   the network the keys refer to after ``prefix`` is removed (``LitModified_UNET.model``, or a sub-module
             with e.g. prefix='model.st_gnn.').
    trainer  optional ``FlatTrainer`` built on ``module``: Adam's exp_avg / exp_avg_sq go into its flat m / v
             buffers, ``step`` into step_count, and the learning rate is set to the scheduler's current value
             (CosineAnnealingLR(T_max) closed form at ``last_epoch``, lit.py:61).
    Returns a dict with what was restored."""
    if isinstance(ckpt, str):
        ckpt = read_checkpoint(ckpt)
    sd = ckpt['state_dict'] if 'state_dict' in ckpt else ckpt
    sd = _strip(sd, prefix)
    own = module.state_dict()
    if strict:
        missing = [k for k in own if k not in sd]
        unexpected = [k for k in sd if k not in own]
        if missing or unexpected:
            raise KeyError(f'checkpoint does not match the module: missing {missing[:5]}, unexpected {unexpected[:5]}')
    with torch.no_grad():            # copy_ keeps parameters that are views of a trainer's flat buffer in place
        for k, t in own.items():
            if k in sd:
                if tuple(t.shape) != tuple(sd[k].shape):
                    raise ValueError(f'{k}: checkpoint shape {tuple(sd[k].shape)} != {tuple(t.shape)}')
                t.copy_(sd[k].to(t.device, t.dtype))
    info = {'tensors': len(sd), 'optimizer': False, 'epoch': ckpt.get('epoch') if isinstance(ckpt, dict) else None}
    if trainer is None or 'optimizer_states' not in ckpt or not ckpt['optimizer_states']:
        return info
    osd = ckpt['optimizer_states'][0]
    names = [k for k, _ in module.named_parameters()]          # Adam indexes parameters() order (lit.py:60)
    ids = [i for grp in osd['param_groups'] for i in grp['params']]
    if len(ids) != len(names):
        raise ValueError(f'optimizer state covers {len(ids)} parameters, the module has {len(names)}')
    steps = set()
    own_params = dict(module.named_parameters())
    trainer.m.zero_()
    trainer.v.zero_()
    for i, k in zip(ids, names):
        st = osd['state'].get(i)
        if st is None:               # parameters that never received a gradient have no Adam state
            continue
        lo, _ = trainer._span[k]
        n = st['exp_avg'].numel()
        want = tuple(own_params[k].shape)
        for key in ('exp_avg', 'exp_avg_sq'):
            if tuple(st[key].shape) != want:
                raise ValueError(f'optimizer state {key} of {k}: shape {tuple(st[key].shape)} != parameter {want}')
        trainer.m[lo:lo + n].copy_(st['exp_avg'].reshape(-1).to(trainer.m.device, torch.float32))
        trainer.v[lo:lo + n].copy_(st['exp_avg_sq'].reshape(-1).to(trainer.v.device, torch.float32))
        steps.add(int(float(st['step'])))
    if len(steps) > 1:
        raise ValueError(f'parameters with different Adam step counts {sorted(steps)}: not a single-group Adam run')
    trainer.step_count = steps.pop() if steps else 0
    grp = osd['param_groups'][0]
    trainer.betas = tuple(grp.get('betas', trainer.betas))
    trainer.eps = grp.get('eps', trainer.eps)
    lr = grp['lr']
    if ckpt.get('lr_schedulers'):
        sch = ckpt['lr_schedulers'][0]
        base = base_lr if base_lr is not None else (sch.get('base_lrs') or [grp.get('initial_lr', lr)])[0]
        lr = cosine_lr(base, sch.get('last_epoch', 0), sch.get('T_max', t_max), sch.get('eta_min', 0.0))
        info['last_epoch'] = sch.get('last_epoch', 0)
    trainer.set_lr(lr)
    info.update(optimizer=True, step=trainer.step_count, lr=lr)
    return info


def lightning_state(module, trainer=None, prefix='model.', epoch=0, base_lr=1e-3, t_max=10):
    """The inverse: a Lightning-shaped checkpoint dict of ``module`` (+ ``trainer``'s Adam state).  What is tested:
    ``state_dict`` loads into the reference's class with ``load_state_dict`` and ``optimizer_states[0]`` into
    ``torch.optim.Adam.load_state_dict`` (tests/test_checkpoint_cpu.py).  Lightning itself is not importable in this image,
    so ``LitModified_UNET.load_from_checkpoint`` on this dict is **parity unpinned**; the keys its checkpoint migration
    indexes (``pytorch-lightning_version``, ``global_step``, ``loops``) are emitted with neutral values so that it has
    what it looks up."""
    out = {'state_dict': {prefix + k: v.detach().clone() for k, v in module.state_dict().items()}, 'epoch': epoch,
           'global_step': int(trainer.step_count) if trainer is not None else 0,
           'pytorch-lightning_version': '2.2.0', 'loops': {}, 'callbacks': {}, 'hparams_name': None}
    if trainer is not None:
        state = {}
        names = [k for k, _ in module.named_parameters()]
        for i, (k, p) in enumerate(module.named_parameters()):
            lo, _ = trainer._span[k]
            n = p.numel()
            state[i] = {'step': torch.tensor(float(trainer.step_count)),
                        'exp_avg': trainer.m[lo:lo + n].view(p.shape).clone(),
                        'exp_avg_sq': trainer.v[lo:lo + n].view(p.shape).clone()}
        out['optimizer_states'] = [{'state': state, 'param_groups': [{
            'lr': trainer.lr, 'betas': tuple(trainer.betas), 'eps': trainer.eps, 'weight_decay': 0, 'amsgrad': False,
            'initial_lr': base_lr, 'params': list(range(len(names)))}]}]
        out['lr_schedulers'] = [{'T_max': t_max, 'eta_min': 0.0, 'base_lrs': [base_lr], 'last_epoch': epoch}]
    return out
