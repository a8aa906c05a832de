"""Host-side graph preparation of the throughput mode (no GPU): node clustering and the block-union CSR form."""
import numpy as np

from oracle import gwnet_ref
from oracle import params as OP
from multimodal_outage_amd.gwnet_engine import BLK_R, BLK_UMAX, block_unions, cluster_order, csr_from_dense


def test_cluster_order_is_a_permutation_and_shrinks_unions():
    N = 3000
    A = gwnet_ref.asym_adj(OP.knn_graph(N, seed=0))
    order = cluster_order([A, A.T])
    assert sorted(order.tolist()) == list(range(N))

    def mean_union(a):
        rowptr, cols, _ = csr_from_dense(a)
        _, uptr, _, umax = block_unions(rowptr, cols, N)
        return float(np.mean(np.diff(uptr))), umax

    before, _ = mean_union(A)
    after, umax = mean_union(A[np.ix_(order, order)])
    assert after < 0.5 * before and umax <= BLK_UMAX, (before, after, umax)


def test_block_unions_reconstruct_the_columns():
    N = 77
    rng = np.random.RandomState(3)
    A = (rng.rand(N, N) < 0.08) * rng.rand(N, N)
    A[5] = 0.0                                              # an empty row
    rowptr, cols, vals = csr_from_dense(A.astype(np.float32))
    lcol, uptr, usrc, umax = block_unions(rowptr, cols, N)
    assert len(uptr) == (N + BLK_R - 1) // BLK_R + 1 and umax == int(np.max(np.diff(uptr)))
    for r in range(N):
        b = r // BLK_R
        got = usrc[uptr[b] + lcol[rowptr[r]:rowptr[r + 1]]]
        assert np.array_equal(got, cols[rowptr[r]:rowptr[r + 1]])
    for b in range(len(uptr) - 1):
        u = usrc[uptr[b]:uptr[b + 1]]
        assert np.all(np.diff(u) > 0)                       # sorted, distinct


def test_disconnected_graph_and_isolated_nodes():
    N = 50
    A = np.zeros((N, N), dtype=np.float32)
    A[0, 1] = A[1, 0] = 1.0
    A[10, 20] = 1.0
    order = cluster_order([A])
    assert sorted(order.tolist()) == list(range(N))


def test_unet_algorithmic_bytes_reproduce_the_survey():
    """bench.unet_alg_bytes_fwd (per-op in+out accounting of the UNet conv stack at the element sizes the engine stores)
    gives SURVEY.md 8(d)'s figures when everything is fp32: 29.7 MB for a 13x256x256 tile, 5.85 MB for 1x128x128; the
    bf16 mode stores fewer bytes, never more."""
    import bench
    assert abs(bench.unet_alg_bytes_fwd(13, 256, 'f32') / 1e6 - 29.7) < 0.05
    assert abs(bench.unet_alg_bytes_fwd(1, 128, 'f32') / 1e6 - 5.85) < 0.01
    b = bench.unet_alg_bytes_fwd(13, 256, 'bf16')
    assert 0.5 * 29.7e6 < b < 29.7e6


def test_plumbing_cache_notices_replaced_tensors():
    """the cached module walk (name -> Parameter, BatchNorm buffers) must not go stale when tensors are replaced
    behind the root module's back -- child._apply alone (torch replaces buffer objects), load_state_dict(assign=True),
    a swapped sub-module, a newly registered parameter."""
    import torch
    import torch.nn as nn
    from multimodal_outage_amd.models.unet import Modified_UNET
    m = Modified_UNET('gwnet', 2, 1, 1)
    named, enc, dec, bufs = m._plumbing()
    ref = dict(m.named_parameters())
    assert set(named) == {k for k in ref if not k.startswith('st_gnn.')} and all(named[k] is ref[k] for k in named)
    assert enc == [k for k in ref if k.split('.')[0] in ('contraction', 'encoder')]
    assert dec == [k for k in ref if k.split('.')[0] in ('decoder', 'expansion')]
    c0 = m.__dict__['_mo_plumbing']
    assert m._plumbing()[0] is named and m.__dict__['_mo_plumbing'] is c0          # unchanged tree: same cache
    m.contraction.double(); m.contraction.float()                                    # child alone: new buffer objects
    bn = m.contraction.inc.double_conv[1]
    assert m._plumbing()[3]['contraction.inc.double_conv.1'][0] is bn.running_mean
    m.load_state_dict({k: v.clone() for k, v in m.state_dict().items()}, assign=True)
    assert m._plumbing()[0]['encoder.fc1.weight'] is m.encoder.fc1.weight
    m.expansion.outc.conv = nn.Conv2d(4, 1, kernel_size=1)
    assert m._plumbing()[0]['expansion.outc.conv.weight'] is m.expansion.outc.conv.weight
    g = m.st_gnn
    from multimodal_outage_amd.models._cache import tree_cache
    c = tree_cache(g, '_mo_named')
    g.bn[0] = nn.BatchNorm2d(32)
    c2 = tree_cache(g, '_mo_named')
    assert c2 is not c and c2.bn['bn.0'][0] is g.bn[0].running_mean
