"""Adjacency helpers of the product: asym_adj (mirror of reference utils.py:152-158), load_adj, the
synthetic k-NN county-like graph of the benchmark and CSR construction."""
import numpy as np

from .gwnet_engine import csr_from_dense  # noqa: F401  (re-export)


def asym_adj(adj):
    """utils.py:152-158: D^-1 A (row-normalised transition matrix), inf -> 0, float32 dense."""
    adj = np.asarray(adj)
    rowsum = np.asarray(adj.sum(1)).flatten()
    with np.errstate(divide='ignore'):
        d_inv = np.power(rowsum, -1).flatten()
    d_inv[np.isinf(d_inv)] = 0.
    return (d_inv[:, None] * adj).astype(np.float32)


def knn_graph(n, mean_degree=6, seed=0):
    """Symmetric k-NN graph of n uniform points in the unit square (RandomState(seed)), 0/1 values,
    zero diagonal, mean degree ~ mean_degree: the structure class of data/graph/adj_mx_fl.csv."""
    rs = np.random.RandomState(seed)
    pts = rs.uniform(size=(n, 2))
    k = max(1, int(round(mean_degree * 0.82)))
    A = np.zeros((n, n), dtype=np.float32)
    for s in range(0, n, 512):
        d = ((pts[s:s + 512, None, :] - pts[None, :, :]) ** 2).sum(-1)
        d[np.arange(d.shape[0]), np.arange(s, s + d.shape[0])] = np.inf
        nb = np.argpartition(d, k, axis=1)[:, :k]
        for r in range(d.shape[0]):
            A[s + r, nb[r]] = 1.0
    A = np.maximum(A, A.T)
    np.fill_diagonal(A, 0.0)
    return A
