"""Shared test helpers: seeded inputs matching tools/make_goldens.py, golden loading, compare."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def rand(seed, shape):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32))


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def assert_close(a, b, atol=1e-4, rtol=1e-4, what=''):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if torch.is_tensor(b) else b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    if not (err <= tol).all():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f'{what}: max|d|={err.max():.3e} at {i}: got {a[i]:.6e} want {b[i]:.6e} '
                             f'(atol={atol}, rtol={rtol})')


def check_grads(named_grads, G, atol=1e-4, rtol=1e-3, prefix=''):
    """named_grads: dict name -> grad tensor (or None).  G: golden npz with grad/ gsample/ gnorm/."""
    none = set(str(s) for s in G['none_grads'])
    for k, g in named_grads.items():
        if k in none:
            assert g is None or float(g.abs().max()) == 0.0, f'{k} should have no grad'
            continue
        assert g is not None, f'{k} missing grad'
        g = g.detach().cpu().numpy()
        if 'grad/' + k in G.files:
            assert_close(g, G['grad/' + k], atol, rtol, prefix + 'grad ' + k)
        else:
            ref = G['gsample/' + k]
            sam = g.reshape(-1)[::max(1, g.size // 2048)][:2048]
            assert_close(sam, ref, atol, rtol, prefix + 'gsample ' + k)
        nrm = np.sqrt((g.astype(np.float64) ** 2).sum())
        ref = float(G['gnorm/' + k])
        assert abs(nrm - ref) <= 1e-3 * max(ref, 1e-3) + 1e-5, (k, nrm, ref)
