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


def check_grads(named_grads, G, atol=1e-4, rtol=1e-3, prefix='', scale_rel=None):
    """named_grads: dict name -> grad tensor (or None).  G: golden npz with grad/ gsample/ gnorm/.
    scale_rel: if set, the absolute tolerance of each tensor is scale_rel * max|ref| (for gradients that
    are sums over millions of positions, where near-zero entries carry the rounding noise of the sum)."""
    none = set(str(s) for s in G['none_grads'])
    for k, g in named_grads.items():
        if k in none:
            assert g is None or float(g.abs().max()) == 0.0, f'{k} should have no grad'
            continue
        assert g is not None, f'{k} missing grad'
        g = g.detach().cpu().numpy()
        if 'grad/' + k in G.files:
            ref = G['grad/' + k]
            a = atol if scale_rel is None else max(atol, scale_rel * float(np.abs(ref).max()))
            assert_close(g, ref, a, rtol, prefix + 'grad ' + k)
        else:
            ref = G['gsample/' + k]
            sam = g.reshape(-1)[::max(1, g.size // 2048)][:2048]
            a = atol if scale_rel is None else max(atol, scale_rel * float(np.abs(ref).max()))
            assert_close(sam, ref, a, rtol, prefix + 'gsample ' + k)
        nrm = np.sqrt((g.astype(np.float64) ** 2).sum())
        ref = float(G['gnorm/' + k])
        assert abs(nrm - ref) <= 1e-3 * max(ref, 1e-3) + 1e-5, (k, nrm, ref)


def check_grads_vs_f64(named_grads, G, factor=3.0, floor=2e-4):
    """Full-model gradients are sums over millions of pixels with heavy cancellation, so the fp32 CPU
    reference itself is only accurate to ~5e-3 of a tensor's scale.  The golden file therefore also holds
    the same step run in float64 (same reference class bodies); the GPU result must be no further from that
    than `factor` x the fp32 CPU reference's own distance (+ `floor`), both relative to max|grad64|."""
    none = set(str(s) for s in G['none_grads'])
    worst = (0.0, None)
    for k, g in named_grads.items():
        if k in none or ('grad64/' + k not in G.files and 'gsample64/' + k not in G.files):
            continue
        g = g.detach().cpu().numpy().astype(np.float64)
        if 'grad64/' + k in G.files:
            r64, r32, got = G['grad64/' + k], G['grad/' + k].astype(np.float64), g
        else:
            r64, r32 = G['gsample64/' + k], G['gsample/' + k].astype(np.float64)
            got = g.reshape(-1)[::max(1, g.size // 2048)][:2048]
        scale = float(np.abs(r64).max())
        if scale < 1e-7:           # mathematically zero gradients (biases in front of a BatchNorm)
            assert float(np.abs(got).max()) < 1e-5, k
            continue
        e_gpu = float(np.abs(got - r64).max()) / scale
        e_cpu = float(np.abs(r32 - r64).max()) / scale
        assert e_gpu <= factor * e_cpu + floor, f'{k}: gpu-vs-f64 {e_gpu:.2e}, cpu32-vs-f64 {e_cpu:.2e}'
        if e_gpu > worst[0]:
            worst = (e_gpu, k)
    return worst


def dropout_keep_mask(seed, thresh, n):
    """Host restatement of the engine's counter-based dropout mask (csrc/mo_common.h: mo_hash32): element idx is kept
    iff hash(seed, idx) >= thresh.  Returns a bool array of n elements."""
    M = 0xFFFFFFFF
    x = (np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B1) + np.uint64(seed)) & np.uint64(M)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & np.uint64(M)
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & np.uint64(M)
    x ^= x >> np.uint64(16)
    x = (x + np.uint64((seed * 0x85ebca6b) & M)) & np.uint64(M)
    x ^= x >> np.uint64(13); x = (x * np.uint64(0xc2b2ae35)) & np.uint64(M)
    x ^= x >> np.uint64(16)
    return x >= np.uint64(thresh)
