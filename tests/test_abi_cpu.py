"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every
symbol include/mo_hip.h declares; the ctypes table mirrors the header; host logic (CSR build, schemas)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, 'include', 'mo_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(mo_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    import multimodal_outage_amd._lib as L
    lib = L.load()
    names = _header_functions()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f'{n} declared in mo_hip.h but not exported by libmo_hip.so'
    assert set(L.SIGNATURES) == set(names), set(L.SIGNATURES) ^ set(names)
    hdr = open(os.path.join(ROOT, 'include', 'mo_hip.h')).read()
    assert lib.mo_version() == L.ABI_VERSION == int(re.search(r'#define\s+MO_ABI_VERSION\s+(\d+)', hdr).group(1))
    assert lib.mo_strerror(-1) == b'invalid argument'
    # argument validation happens before any device work: NULL pointers are rejected on a CPU-only host
    assert lib.mo_spmm_csr(None, None, None, 0, None, None, 0, 0, 0, 0, None) == -1
    assert lib.mo_conv1x1_fwd(None, 0, 0, 0, 0, 0, None, None, 0, None, 0, 0, 0, None) == -1


def test_product_has_no_cpu_fallback():
    from multimodal_outage_amd.models.graph_wavenet import gwnet
    m = gwnet('cpu', num_nodes=5, in_dim=2, out_dim=2)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 2, 5, 3))
    # the product never imports the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'multimodal_outage_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), f


def test_csr_bit_exact_vs_scipy_and_reference_graph():
    import scipy.sparse as sp
    from multimodal_outage_amd.graphs import csr_from_dense, asym_adj, knn_graph
    G = np.load(os.path.join(ROOT, 'tests', 'golden', 'adjacency.npz'))
    A = G['adj']
    rp, ci, va = csr_from_dense(A)
    assert (rp == G['rowptr']).all() and (ci == G['colidx']).all() and (va == G['vals']).all()
    rp, ci, va = csr_from_dense(A.T)
    assert (rp == G['t_rowptr']).all() and (ci == G['t_colidx']).all()
    assert (asym_adj(A) == G['asym']).all()
    K = knn_graph(300, seed=2)
    c = sp.csr_matrix(asym_adj(K))
    rp, ci, va = csr_from_dense(asym_adj(K))
    assert (rp == c.indptr).all() and (ci == c.indices).all() and (va == c.data).all()


def test_state_dict_schema_matches_reference():
    from oracle import params as P
    from multimodal_outage_amd.models.graph_wavenet import gwnet, load_adj
    from multimodal_outage_amd.models.unet import Modified_UNET, UNet
    m = Modified_UNET('gwnet', 7, 1, 1)
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == \
        [(k, tuple(v)) for k, v in P.unet_schema().items()]
    assert UNet is Modified_UNET
    g = gwnet('cpu', num_nodes=20, supports=[np.eye(20), np.eye(20)], in_dim=2, out_dim=12, kernel_size=2)
    assert g.receptive_field == 13 and g.supports_len == 3
    assert [(k, tuple(v.shape)) for k, v in g.state_dict().items()] == \
        [(k, tuple(v)) for k, v in P.gwnet_schema(num_nodes=20, supports_len=3, in_dim=2, out_dim=12,
                                                    kernel_size=2).items()]
    # (RNG consumption order of the constructor vs the reference's: tests/test_checkpoint_cpu.py, against the
    # reference class's own initial values)


def test_aptinit_svd_initialisation():
    """graph_wavenet.py:138-142: nodevec1/2 from the rank-10 SVD of aptinit."""
    from multimodal_outage_amd.models.graph_wavenet import gwnet
    torch.manual_seed(3)
    A = torch.rand(30, 30)
    g = gwnet('cpu', num_nodes=30, supports=[np.eye(30)], aptinit=A, in_dim=2, out_dim=2)
    m, p, n = torch.svd(A)
    e1 = torch.mm(m[:, :10], torch.diag(p[:10] ** 0.5))
    e2 = torch.mm(torch.diag(p[:10] ** 0.5), n[:, :10].t())
    assert torch.allclose(g.nodevec1, e1) and torch.allclose(g.nodevec2, e2)
    assert g.supports_len == 2 and list(g.state_dict())[:2] == ['nodevec1', 'nodevec2']


def test_every_call_site_matches_the_ctypes_table():
    """Walk every `L.call('mo_*', ...)` / `lib.mo_*(...)` of the package (and bench.py) with ast and check the number of
    arguments against _lib.SIGNATURES -- round 2 shipped functional.py with 7 arguments for the 8-argument
    mo_nchw_to_nbtc (an ABI change the call site did not follow; nothing ran it)."""
    import ast
    import multimodal_outage_amd._lib as L
    files = [os.path.join(ROOT, 'bench.py'), os.path.join(ROOT, '__graft_entry__.py')]
    for dirpath, _, fs in os.walk(os.path.join(ROOT, 'multimodal_outage_amd')):
        files += [os.path.join(dirpath, f) for f in fs if f.endswith('.py')]
    checked = 0
    for path in files:
        tree = ast.parse(open(path).read(), path)
        for node in ast.walk(tree):
            if not isinstance(node, ast.Call):
                continue
            f = node.func
            name = nargs = None
            if (isinstance(f, ast.Attribute) and f.attr == 'call' and node.args
                    and isinstance(node.args[0], ast.Constant) and isinstance(node.args[0].value, str)
                    and node.args[0].value.startswith('mo_')):
                name, args = node.args[0].value, node.args[1:]
            elif isinstance(f, ast.Attribute) and f.attr.startswith('mo_') and f.attr in L.SIGNATURES:
                name, args = f.attr, node.args
            if name is None:
                continue
            assert name in L.SIGNATURES, f'{path}:{node.lineno}: unknown entry point {name}'
            want = len(L.SIGNATURES[name][1])
            # `*view.args()` (6 values) and `*_NOVIEW` (6) are the two starred forms of the UNet engine; `*drop` is 3
            n = 0
            for a in args:
                if isinstance(a, ast.Starred):
                    src = ast.unparse(a.value)
                    n += 6 if ('args()' in src or '_NOVIEW' in src or src in ('a1',)) else None
                else:
                    n += 1
            assert n == want, f'{path}:{node.lineno}: {name} called with {n} arguments, the ABI takes {want}'
            checked += 1
    assert checked >= 120, checked
