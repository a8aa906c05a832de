"""Host-side orchestration of the Graph-WaveNet forward/backward on the HIP C-ABI.

One ``torch.autograd.Function`` spans the whole network body (graph_wavenet.py:191-254); torch only
owns the buffers and carries the gradients to the parameter tensors.  Internal activations live in
the node-major channels-last "nbtc" layout (rows p=(n*B+b)*T+t of 32 channels), so every node-axis
product is a row operation on [N][B*T*32] and every channel contraction a GEMM over rows.

BatchNorm of layer i is never materialised: the kernels that consume x_{i+1} = BN(h_i) apply the
folded affine (scale, shift) on load (tcn / residual), and the backward recomputes xhat from h_i.
"""
import ctypes as _C
import numpy as np
import os

import torch

from . import _lib as L


def csr_from_dense(a):
    """Row-major CSR (rowptr, colidx int32; vals float32) of a dense matrix: the same ordering as
    scipy.sparse.csr_matrix(a) (rows ascending, columns ascending within a row)."""
    import ctypes as C
    a = np.ascontiguousarray(a, dtype=np.float32)
    n, m = a.shape
    rowptr = np.zeros(n + 1, dtype=np.int32)
    nnz = C.c_long(0)
    L.call('mo_csr_from_dense', a.ctypes.data, n, m, rowptr.ctypes.data, None, None, C.byref(nnz))
    cols = np.zeros(max(nnz.value, 1), dtype=np.int32)
    vals = np.zeros(max(nnz.value, 1), dtype=np.float32)
    L.call('mo_csr_from_dense', a.ctypes.data, n, m, rowptr.ctypes.data, cols.ctypes.data, vals.ctypes.data,
           C.byref(nnz))
    return rowptr, cols[:nnz.value], vals[:nnz.value]


BLK_R, BLK_UMAX = 16, 64     # mo_spmm_blk: rows per block, largest neighbour union it stages (include/mo_hip.h)
BLK_MIN_J = 16384            # below this row length the plain CSR kernel is as fast (measured, tools/bench_spmm.py)
SPMM_DUAL = os.environ.get('MO_SPMM_DUAL', '1') != '0'    # A/B switch: two supports' accumulations into dg as one launch
G_BF_ONLY = os.environ.get('MO_G_BF_ONLY', '1') != '0'    # A/B switch: throughput mode keeps g as bf16 + the skip crop in fp32


def cluster_order(patterns, R=BLK_R):
    """Node renumbering for the blocked SpMM: order[new] = old.  Greedy clusters of R nodes grown breadth-first over
    the union pattern of all supports (seeds in global BFS order), so that R consecutive rows share most neighbours
    (k-NN graph of the benchmark: 36.9 distinct sources per 16 rows instead of 94.8 in file order)."""
    pat = None
    for a in patterns:
        a = np.asarray(a) != 0
        pat = (a | a.T) if pat is None else (pat | a | a.T)
    n = pat.shape[0]
    rows, cols = np.nonzero(pat)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr)
    seen = np.zeros(n, dtype=bool)
    bfs = []
    for s0 in range(n):
        if seen[s0]:
            continue
        seen[s0] = True
        q, qi = [s0], 0
        while qi < len(q):
            v = q[qi]; qi += 1
            for u in cols[rowptr[v]:rowptr[v + 1]]:
                if not seen[u]:
                    seen[u] = True
                    q.append(int(u))
        bfs.extend(q)
    taken = np.zeros(n, dtype=bool)
    order = []
    for seed in bfs:
        if taken[seed]:
            continue
        taken[seed] = True
        cl, qi = [seed], 0
        while len(cl) < R and qi < len(cl):
            v = cl[qi]; qi += 1
            for u in cols[rowptr[v]:rowptr[v + 1]]:
                if not taken[u]:
                    taken[u] = True
                    cl.append(int(u))
                    if len(cl) >= R:
                        break
        order.extend(cl)
    return np.asarray(order, dtype=np.int64)


def block_unions(rowptr, cols, n, R=BLK_R):
    """Per block of R rows: the sorted distinct columns (usrc, uptr) and every entry's position in them (lcol)."""
    nb = (n + R - 1) // R
    uptr = np.zeros(nb + 1, dtype=np.int32)
    usrc, lcol = [], np.zeros(len(cols), dtype=np.int32)
    for b in range(nb):
        lo, hi = int(rowptr[b * R]), int(rowptr[min(n, (b + 1) * R)])
        u, inv = np.unique(cols[lo:hi], return_inverse=True)
        usrc.append(u.astype(np.int32))
        lcol[lo:hi] = inv
        uptr[b + 1] = uptr[b] + len(u)
    usrc = np.concatenate(usrc) if usrc else np.zeros(0, dtype=np.int32)
    if len(usrc) == 0:
        usrc = np.zeros(1, dtype=np.int32)
    return lcol, uptr, usrc, int(np.max(np.diff(uptr))) if nb else 0


class StaticSupport:
    """A static support A (N,N): CSR of A^T for the forward nconv (out[w] = sum_v A[v,w] x[v],
    graph_wavenet.py:65) and CSR of A for its backward.  With `order` (new -> old node) the matrix is renumbered
    first and each CSR also carries its block-union form for mo_spmm_blk (None when a union exceeds BLK_UMAX)."""

    def __init__(self, dense, device, order=None):
        dense = np.asarray(dense, dtype=np.float32)
        if order is not None:
            dense = dense[np.ix_(order, order)]
        self.n = dense.shape[0]
        self.fwd = self._pack(dense.T, device, order is not None)
        self.bwd = self._pack(dense, device, order is not None)
        self.nnz = int(self.fwd[1].numel())

    def _pack(self, a, device, blocked):
        rowptr, cols, vals = csr_from_dense(a)
        # the kernels trust these arrays (a gather through a bad column index is a device memory fault, not an MO_E*
        # return -- round 1's only GPU abort was a test helper handing the C-ABI index arrays that had been freed):
        # validate once on the host, and keep the device copies alive as attributes of this object
        if not (rowptr[0] == 0 and (np.diff(rowptr) >= 0).all() and rowptr[-1] == len(cols)
                and (len(cols) == 0 or (cols.min() >= 0 and cols.max() < self.n))):
            raise ValueError('StaticSupport: malformed CSR')
        blk = None
        if blocked:
            lcol, uptr, usrc, umax = block_unions(rowptr, cols, self.n)
            if umax <= BLK_UMAX:
                blk = tuple(torch.from_numpy(x).to(device) for x in (lcol, uptr, usrc)) + (umax,)
        return tuple(torch.from_numpy(x).to(device) for x in (rowptr, cols, vals)) + (blk,)


def _e(n, w, dev):
    return torch.empty((n, w), device=dev, dtype=torch.float32)


class GwnetConfig:
    def __init__(self, *, num_nodes, in_dim, out_dim, kernel_size, blocks, layers, skip_channels,
                 end_channels, gcn, adaptive, dropout, names):
        self.N, self.Cin, self.Cout, self.K = num_nodes, in_dim, out_dim, kernel_size
        self.blocks, self.layers = blocks, layers
        self.L = blocks * layers
        self.dil = [2 ** i for _ in range(blocks) for i in range(layers)]
        self.rf = 1 + sum((kernel_size - 1) * d for d in self.dil)
        self.Cs, self.Ce = skip_channels, end_channels
        self.gcn, self.adaptive, self.dropout = gcn, adaptive, dropout
        self.names = names            # parameter order of the autograd Function
        self.grad_out = None          # optional {name: preallocated grad tensor} (flat-buffer trainer)
        self.grad_ready = None        # optional callback(names): these gradients are final (overlapped all-reduce)
        self.dense_bf16 = False       # bf16 operands (fp32 accumulate) for the dense adaptive products
        self.overlap = True           # dense branch on a side stream beside the sparse branch


# Optional live kernel timing (bench.py roofline leg): when PROFILE is a list, every dense
# node-axis product is bracketed by HIP events on the launching stream and appended as
# (name, algorithmic_flops, start_event, end_event).
PROFILE = None
SERIAL = bool(int(os.environ.get('MO_SERIAL', '0')))   # True: disable the side-stream overlap (bench.py's
                                                         # un-contended roofline pass; MO_SERIAL=1 for profiling)


def _dense(name, N, J, *args):
    flops = 2.0 * N * N * J
    if PROFILE is None:
        L.call(name, *args)
        return
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    L.call(name, *args)
    e1.record()
    PROFILE.append((name, flops, e0, e1))


def _bf16(x):
    """fp32 -> bf16 copy (round to nearest even) for the bf16-operand dense products."""
    y = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    L.call('mo_f32_to_bf16', L.ptr(x), L.ptr(y), x.numel(), L.stream())
    return y


BIG_TILE_MIN_N = 1024   # node count from which the 256x256 LDS-DMA-ring kernel is used


def _bf16_padded(x):
    """(N,N) fp32 -> bf16 [N][Kpad] with zero columns beyond N (Kpad = N rounded up to 32)."""
    n, k = x.shape
    kpad = (k + 31) // 32 * 32
    y = torch.empty((n, kpad), device=x.device, dtype=torch.bfloat16)
    L.call('mo_f32_to_bf16_padded', L.ptr(x), n, k, L.ptr(y), kpad, L.stream())
    return y


def _use_ring(M, Ncols):
    """Pick the 256x256 LDS-DMA-ring kernel (one workgroup per CU, ~1.07 PF per full wave of 256 tiles) over
    the 128x128 kernel (4 workgroups per CU, ~0.7 PF, tolerant of ragged grids) from the tile-grid fill;
    rates measured with tools/bench_gemm.py on MI355X."""
    if M < BIG_TILE_MIN_N:
        return False
    tiles = ((M + 255) // 256) * ((Ncols + 255) // 256)
    fill = tiles / (((tiles + 255) // 256) * 256)
    est128 = 700.0 if Ncols >= 4096 else 190.0 * Ncols / 1024.0
    return fill * 1070.0 > est128


_SKIP = os.environ.get('MO_SKIP', '')     # timing experiments only: drop a kernel class from the step
# MO_OVL_DENSE=0: the dense products of the data path stay on the main stream (sparse branch first, then the dense
# one); only the dA accumulations and the weight-gradient lane run beside the main chain
OVL_DENSE = bool(int(os.environ.get('MO_OVL_DENSE', '0')))
OVL_DA = bool(int(os.environ.get('MO_OVL_DA', '1')))      # 0: the dA accumulations stay on the main stream too


def _adj_prod(A_bf, X_bf, Y, N, J, beta, Y_bf=None):
    """Y[N][J] (+)= A_bf[N][N(k)] @ X_bf[N(k)][J]   (bf16 operands, fp32 accumulate); A_bf is [N][Kpad].
    Y_bf: optional bf16 copy of the result written by the same epilogue."""
    if 'prod' in _SKIP:
        return
    kpad = A_bf.shape[1]
    if _use_ring(N, J):
        _dense('mo_gemm_bf16_256', N, J, L.ptr(A_bf), kpad, kpad, L.ptr(X_bf), J, 1, L.ptr(Y), J, N, J, N, beta,
               L.ptr(Y_bf), L.stream())
    else:
        _dense('mo_gemm_bf16', N, J, L.ptr(A_bf), kpad, L.ptr(X_bf), J, 1, L.ptr(Y), J, N, J, N, beta, L.ptr(Y_bf),
               L.stream())


def _adj_grad_bf(X_bf, dY_bf, dA, N, J, beta):
    """dA[N][N] (+)= X_bf[N][J] @ dY_bf[N][J]^T."""
    if 'dA' in _SKIP:
        return
    if N >= BIG_TILE_MIN_N:
        _dense('mo_gemm_bf16_256', N, J, L.ptr(X_bf), J, J, L.ptr(dY_bf), J, 0, L.ptr(dA), N, N, N, J, beta, None,
               L.stream())
    else:
        _dense('mo_gemm_bf16', N, J, L.ptr(X_bf), J, L.ptr(dY_bf), J, 0, L.ptr(dA), N, N, N, J, beta, None,
               L.stream())


_SIDE = {}


def _side_stream(dev):
    """Side HIP streams (one per device and role): the dense adaptive-adjacency branch, the weight gradients."""
    role = 'dense'
    if isinstance(dev, tuple):
        dev, role = dev
    key = (dev.type, dev.index, role)
    if key not in _SIDE:
        # the dense branch holds one 128-KB-LDS workgroup per CU: give it dispatch priority over the
        # HBM-bound kernels it runs beside, or it starves behind their many small workgroups
        _SIDE[key] = torch.cuda.Stream(device=dev, priority=-1 if role == 'dense' else 0)
    return _SIDE[key]


class _WgradLane:
    """Weight/bias-gradient kernels are off the data-flow critical path of the backward pass: they run on a
    second side stream beside the next kernels of the main chain and are joined once at the end.  Tensors
    allocated on the main stream and read here are protected with record_stream; workspaces are allocated
    under the side stream."""

    def __init__(self, dev, enabled):
        self.main = torch.cuda.current_stream()
        self.side = _side_stream((dev, 'wgrad')) if enabled else None

    def run(self, fn, reads=()):
        if 'wgrad' in _SKIP:
            return
        if self.side is None:
            fn()
            return
        self.side.wait_stream(self.main)
        for t in reads:
            if t is not None:
                t.record_stream(self.side)
        with torch.cuda.stream(self.side):
            fn()

    def join(self):
        if self.side is not None:
            self.main.wait_stream(self.side)


def _spmm(csr, n, X, Y, J, beta):
    """Y (+)= S @ X over nbtc rows; the storage type of each side (fp32 / bf16) follows the tensors."""
    blk = csr[3]
    if blk is not None and X.dtype == torch.bfloat16 and J >= BLK_MIN_J:
        L.call('mo_spmm_blk', L.ptr(csr[0]), L.ptr(blk[0]), L.ptr(csr[2]), L.ptr(blk[1]), L.ptr(blk[2]), n, blk[3],
               L.ptr(X), L.ptr(Y), J, beta, int(Y.dtype == torch.bfloat16), L.stream())
        return
    L.call('mo_spmm_csr', L.ptr(csr[0]), L.ptr(csr[1]), L.ptr(csr[2]), n, L.ptr(X), L.ptr(Y), J, beta,
           int(X.dtype == torch.bfloat16), int(Y.dtype == torch.bfloat16), L.stream())


def _spmm2_ok(a, b, Y, J):
    (ca, xa), (cb, xb) = a, b
    return bool(SPMM_DUAL and ca[3] is not None and cb[3] is not None and xa.dtype == torch.bfloat16 and
                xb.dtype == torch.bfloat16 and J >= BLK_MIN_J)


def _spmm2(ca, xa, cb, xb, n, Y, J, beta):
    ba, bb = ca[3], cb[3]
    L.call('mo_spmm_blk2', L.ptr(ca[0]), L.ptr(ba[0]), L.ptr(ca[2]), L.ptr(ba[1]), L.ptr(ba[2]), ba[3], L.ptr(xa),
           L.ptr(cb[0]), L.ptr(bb[0]), L.ptr(cb[2]), L.ptr(bb[1]), L.ptr(bb[2]), bb[3], L.ptr(xb),
           n, L.ptr(Y), J, beta, int(Y.dtype == torch.bfloat16), L.stream())


def _bf_mask(ts):
    m = 0
    for k, t in enumerate(ts):
        if t.dtype == torch.bfloat16:
            m |= 1 << k
    return m


class GwnetFunction(torch.autograd.Function):
    """y = gwnet_body(x; params).  x: (B,Cin,N,T) fp32 contiguous on the GPU."""

    @staticmethod
    def forward(ctx, cfg, statics, bn_bufs, training, node_new, x, *params):
        p = dict(zip(cfg.names, params))
        dev = x.device
        st = L.stream()
        B, Cin, N, T = x.shape
        assert Cin == cfg.Cin and N == cfg.N
        x = x.contiguous()
        G = N * B
        K = cfg.K
        Tp = max(T, cfg.rf)
        pad = Tp - T
        Tf = Tp - (cfg.rf - 1)
        saved = {}

        x_int = _e(G * T, Cin, dev)
        # node_new (int32 [N] or None): the renumbering of the node axis is folded into the boundary transposes
        L.call('mo_nchw_to_nbtc', L.ptr(x), L.ptr(x_int), B, Cin, N, T, L.ptr(node_new), st)
        h = _e(G * Tp, 32, dev)
        L.call('mo_conv1x1_fwd', L.ptr(x_int), Cin, Tp, T, -pad, 0, L.ptr(p['start_conv.weight']),
               L.ptr(p['start_conv.bias']), 32, L.ptr(h), G * Tp, 0, 0, st)
        adp = adpT = None
        if cfg.gcn and cfg.adaptive:
            adp = _e(N, N, dev)
            adpT = _e(N, N, dev)
            L.call('mo_adp_fwd', L.ptr(p['nodevec1']), L.ptr(p['nodevec2']), N, p['nodevec1'].shape[1],
                   L.ptr(adp), L.ptr(adpT), st)
        # (more than 3 supports = more than 7 mlp sources: the tile engine serves them, in fp32 storage only)
        use_bf = bool(cfg.dense_bf16 and adp is not None and N % 8 == 0 and 2 * (len(statics) + 1) + 1 <= 7)
        adp_bf = adpT_bf = None
        if use_bf:
            adp_bf, adpT_bf = _bf16_padded(adp), _bf16_padded(adpT)
        skip = _e(G * Tf, cfg.Cs, dev)
        drop_p = cfg.dropout if training else 0.0
        thresh = int(min(max(drop_p, 0.0), 0.999999) * 4294967296.0) if drop_p > 0 else 0
        dscale = 1.0 / (1.0 - drop_p) if drop_p > 0 else 1.0
        base_seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if thresh else 0

        scale = shift = None
        Tin = Tp
        layers = []
        for i in range(cfg.L):
            d = cfg.dil[i]
            Tout = Tin - d * (K - 1)
            P = G * Tout
            J = B * Tout * 32
            Wp = torch.empty(K * 64 * 32, device=dev, dtype=torch.float32)
            L.call('mo_tcn_pack_weights', L.ptr(p[f'filter_convs.{i}.weight']),
                   L.ptr(p[f'gate_convs.{i}.weight']), K, L.ptr(Wp), st)
            g_bf = torch.empty((P, 32), device=dev, dtype=torch.bfloat16) if use_bf else None
            if use_bf and G_BF_ONLY:
                # throughput mode: g exists as its bf16 copy -- what the node-axis products, the mlp and its weight gradient
                # round it to anyway (bit-identical results) -- plus the fp32 values of the last Tf steps, all the skip
                # path reads: the full fp32 tensor (0.64 GB per layer written, read twice) is gone
                g = _e(G * Tf, 32, dev)
                L.call('mo_tcn_fwd', L.ptr(h), L.ptr(scale), L.ptr(shift), L.ptr(Wp),
                       L.ptr(p[f'filter_convs.{i}.bias']), L.ptr(p[f'gate_convs.{i}.bias']), K, d, G, Tin,
                       None, L.ptr(g_bf), 1, L.ptr(g), Tf, st)
                g_T = Tf
                srcs = [g_bf]
            else:
                g = _e(P, 32, dev)
                L.call('mo_tcn_fwd', L.ptr(h), L.ptr(scale), L.ptr(shift), L.ptr(Wp),
                       L.ptr(p[f'filter_convs.{i}.bias']), L.ptr(p[f'gate_convs.{i}.bias']), K, d, G, Tin,
                       L.ptr(g), L.ptr(g_bf), int(use_bf), None, 0, st)
                g_T = Tout
                srcs = [g]
            bf_saved = None
            if cfg.gcn:
                # the dense (adaptive, MFMA-bound) branch runs on a side HIP stream beside the sparse
                # (static CSR, HBM-bound) branch; both only read g and are joined in front of the mlp
                side = _side_stream(dev) if (cfg.adaptive and statics and cfg.overlap and not SERIAL
                                             and OVL_DENSE) else None
                dense_out = []
                # bf16 mode keeps every diffusion intermediate (and, in backward, its gradient) as a bf16
                # tensor only: the row-streaming kernels widen on load and narrow on store
                idt = torch.bfloat16 if use_bf else torch.float32
                if cfg.adaptive:
                    x1 = torch.empty((P, 32), device=dev, dtype=idt)
                    x2 = torch.empty((P, 32), device=dev, dtype=idt)
                    main = torch.cuda.current_stream()
                    if side is not None:
                        side.wait_stream(main)
                    with torch.cuda.stream(side if side is not None else main):
                        if use_bf:
                            _adj_prod(adpT_bf, g_bf, None, N, J, 0, x1)
                            _adj_prod(adpT_bf, x1, None, N, J, 0, x2)
                            bf_saved = (g_bf, x1)
                        else:
                            _dense('mo_adj_gemm', N, J, L.ptr(adp), N, L.ptr(g), L.ptr(x1), J, 0, L.stream())
                            _dense('mo_adj_gemm', N, J, L.ptr(adp), N, L.ptr(x1), L.ptr(x2), J, 0, L.stream())
                    dense_out = [x1, x2]
                for s in statics:
                    x1 = torch.empty((P, 32), device=dev, dtype=idt)
                    x2 = torch.empty((P, 32), device=dev, dtype=idt)
                    _spmm(s.fwd, N, g_bf if use_bf else g, x1, J, 0)     # throughput mode gathers the bf16 copy
                    _spmm(s.fwd, N, x1, x2, J, 0)
                    srcs += [x1, x2]
                srcs += dense_out
                if side is not None:
                    torch.cuda.current_stream().wait_stream(side)
                W, bb = p[f'gconv.{i}.mlp.mlp.weight'], p[f'gconv.{i}.mlp.mlp.bias']
                lt, ls, seed = thresh, dscale, (base_seed + 7919 * i) & 0xFFFFFFFF
            else:
                W, bb = p[f'residual_convs.{i}.weight'], p[f'residual_convs.{i}.bias']
                lt, ls, seed = 0, 1.0, 0
            hn = _e(P, 32, dev)
            nblk = (P + 127) // 128
            partial = torch.empty(L.load().mo_mlp_partial_floats(P), device=dev, dtype=torch.float32)
            L.call('mo_gcn_mlp_fwd', L.ptr_array(srcs), len(srcs), L.ptr(W), L.ptr(bb), G, Tout, Tin,
                   L.ptr(h), L.ptr(scale), L.ptr(shift), seed, lt, ls, L.ptr(hn), L.ptr(partial),
                   _bf_mask(srcs), st)
            stats = torch.empty(4, 32, device=dev, dtype=torch.float32)   # scale, shift, mean, rstd
            rm, rv = bn_bufs[i]
            L.call('mo_bn_finalize', L.ptr(partial), nblk, P, L.ptr(p[f'bn.{i}.weight']),
                   L.ptr(p[f'bn.{i}.bias']), L.ptr(rm), L.ptr(rv), 0.1, 1e-5, 1 if training else 0,
                   L.ptr(stats[0]), L.ptr(stats[1]), L.ptr(stats[2]), L.ptr(stats[3]), st)
            layers.append(dict(h_in=h, scale=scale, shift=shift, Wp=Wp, g=g, g_T=g_T, srcs=srcs, h=hn, bf=bf_saved,
                               stats=stats, Tin=Tin, Tout=Tout, seed=seed, thresh=lt, dscale=ls))
            h, scale, shift, Tin = hn, stats[0], stats[1], Tout

        # skip path (graph_wavenet.py:230-236): only the last Tf steps of every layer's skip conv reach the head,
        # so all of them are ONE contraction over the concatenated channels; skip is written once
        P_f = G * Tf
        # throughput mode: the head's two real contractions (end_conv_1 forward / data gradient) run as bf16 ring-GEMM
        # launches with bias+ReLU / ReLU-gate epilogues; skip is then stored post-ReLU (all its consumers apply ReLU)
        head_bf = bool(use_bf and cfg.L <= 8 and cfg.Cs % 32 == 0 and cfg.Ce % 32 == 0 and cfg.Cout <= 16
                       and 256 % (cfg.Ce // 4) == 0 and P_f < (1 << 31))
        skip_bf = torch.empty((P_f, cfg.Cs), device=dev, dtype=torch.bfloat16) if head_bf else None
        for c0 in range(0, cfg.L, 8):
            ids = list(range(c0, min(cfg.L, c0 + 8)))
            bsum = torch.stack([p[f'skip_convs.{i}.bias'] for i in ids]).sum(0)
            touts = (_C.c_int * len(ids))(*[layers[i]['g_T'] for i in ids])      # (steps per group of the stored fp32 g)
            L.call('mo_skip_fwd', L.ptr_array([layers[i]['g'] for i in ids]), touts,
                   L.ptr_array([p[f'skip_convs.{i}.weight'] for i in ids]), len(ids), L.ptr(bsum), cfg.Cs, G, Tf,
                   L.ptr(skip), 1 if c0 > 0 else 0, int(head_bf), L.ptr(skip_bf), st)
        r1 = _e(P_f, cfg.Ce, dev)
        W1_bf = None
        if head_bf:
            W1 = p['end_conv_1.weight']
            W1_bf = torch.empty((cfg.Ce, cfg.Cs), device=dev, dtype=torch.bfloat16)
            L.call('mo_f32_to_bf16', L.ptr(W1), L.ptr(W1_bf), W1.numel(), st)
            # output-bound (K = Cs against a P_f x Ce fp32 result): the 128x128 kernel, four workgroups per CU
            L.call('mo_gemm_bf16_ex', L.ptr(skip_bf), cfg.Cs, L.ptr(W1_bf), cfg.Cs, 0, L.ptr(r1), cfg.Ce,
                   P_f, cfg.Ce, cfg.Cs, 0, None, L.ptr(p['end_conv_1.bias']), 1, None, st)
        else:
            L.call('mo_conv1x1_fwd', L.ptr(skip), cfg.Cs, 0, 0, 0, 1, L.ptr(p['end_conv_1.weight']),
                   L.ptr(p['end_conv_1.bias']), cfg.Ce, L.ptr(r1), P_f, 1, 0, st)
        y_int = _e(P_f, cfg.Cout, dev)
        L.call('mo_conv1x1_fwd', L.ptr(r1), cfg.Ce, 0, 0, 0, 0, L.ptr(p['end_conv_2.weight']),
               L.ptr(p['end_conv_2.bias']), cfg.Cout, L.ptr(y_int), P_f, 0, 0, st)
        y = torch.empty((B, cfg.Cout, N, Tf), device=dev, dtype=torch.float32)
        L.call('mo_nbtc_to_nchw', L.ptr(y_int), L.ptr(y), B, cfg.Cout, N, Tf, L.ptr(node_new), st)

        ctx.cfg, ctx.statics, ctx.training, ctx.node_new = cfg, statics, training, node_new
        ctx.dims = (B, N, T, Tp, Tf, G)
        ctx.layers, ctx.x_int, ctx.adp, ctx.adpT, ctx.skip, ctx.r1 = layers, x_int, adp, adpT, skip, r1
        ctx.adp_bf = adp_bf
        ctx.mfma_bf16 = int(use_bf)
        ctx.W1_bf, ctx.skip_bf = W1_bf, skip_bf
        ctx.params = p
        ctx.x_needs_grad = x.requires_grad
        return y

    @staticmethod
    def backward(ctx, dy):
        cfg, statics, p = ctx.cfg, ctx.statics, ctx.params
        if not ctx.training:
            raise RuntimeError('gwnet backward is only defined in training mode (batch-stat BatchNorm)')
        lib = L.load()
        B, N, T, Tp, Tf, G = ctx.dims
        K = cfg.K
        dev = dy.device
        st = L.stream()
        grads = {k: None for k in cfg.names}
        gout = cfg.grad_out or {}
        lane = _WgradLane(dev, cfg.overlap and not SERIAL)

        def gbuf(name, like=None, shape=None):
            """Gradient destination: the trainer's flat-buffer view when registered, else a new tensor."""
            t = gout.get(name)
            if t is not None:
                return t
            return torch.empty_like(like) if like is not None else torch.empty(shape, device=dev,
                                                                              dtype=torch.float32)

        def ws_for(M, Nn, P):
            return torch.empty(lib.mo_wgrad_ws_floats(M, Nn, P), device=dev, dtype=torch.float32)

        # ---- head (graph_wavenet.py:252-254)
        P_f = G * Tf
        dy = dy.contiguous()
        dy_int = _e(P_f, cfg.Cout, dev)
        L.call('mo_nchw_to_nbtc', L.ptr(dy), L.ptr(dy_int), B, cfg.Cout, N, Tf, L.ptr(ctx.node_new), st)
        skip, r1 = ctx.skip, ctx.r1
        ws = ws_for(max(cfg.Ce, cfg.Cout), max(cfg.Ce, cfg.Cs), P_f)
        gW2 = gbuf('end_conv_2.weight', p['end_conv_2.weight']); gb2 = gbuf('end_conv_2.bias', p['end_conv_2.bias'])
        lane.run(lambda: L.call('mo_conv1x1_bwd_weight', L.ptr(dy_int), cfg.Cout, P_f, L.ptr(r1), cfg.Ce, 0, 0, 0, 0,
                                L.ptr(gW2), L.ptr(gb2), L.ptr(ws_for(cfg.Cout, cfg.Ce, P_f)), L.stream()),
                 reads=(dy_int,))
        da1 = _e(P_f, cfg.Ce, dev)
        da1_bf = None
        if ctx.W1_bf is not None:
            da1_bf = torch.empty((P_f, cfg.Ce), device=dev, dtype=torch.bfloat16)
            L.call('mo_conv1x1_bwd_data_smallk', L.ptr(dy_int), cfg.Cout, P_f, L.ptr(p['end_conv_2.weight']), cfg.Ce,
                   L.ptr(r1), L.ptr(da1), L.ptr(da1_bf), st)
        else:
            L.call('mo_conv1x1_bwd_data', L.ptr(dy_int), cfg.Cout, P_f, L.ptr(p['end_conv_2.weight']), cfg.Ce,
                   L.ptr(da1), 0, 0, 0, L.ptr(r1), 0, st)
        gW1 = gbuf('end_conv_1.weight', p['end_conv_1.weight']); gb1 = gbuf('end_conv_1.bias', p['end_conv_1.bias'])
        if da1_bf is not None and ctx.skip_bf is not None and P_f % 32 == 0 and cfg.Ce % 8 == 0 and cfg.Cs % 8 == 0:
            # both operands k-major bf16 as they lie: split-K ring GEMM; the bias gradient is an fp32 column sum
            def _w1():
                wsk = torch.empty(lib.mo_wgrad_bf16_kk_ws_floats(cfg.Ce, cfg.Cs, P_f), device=dev, dtype=torch.float32)
                L.call('mo_wgrad_bf16_kk', L.ptr(da1_bf), cfg.Ce, L.ptr(ctx.skip_bf), cfg.Cs, P_f, cfg.Ce, cfg.Cs,
                       L.ptr(gW1), L.ptr(wsk), L.stream())
                wsc = torch.empty(lib.mo_colsum_ws_floats(P_f, cfg.Ce), device=dev, dtype=torch.float32)
                L.call('mo_colsum', L.ptr(da1), P_f, cfg.Ce, L.ptr(gb1), L.ptr(wsc), L.stream())
            lane.run(_w1, reads=(da1, da1_bf))
        else:
            lane.run(lambda: L.call('mo_conv1x1_bwd_weight', L.ptr(da1), cfg.Ce, P_f, L.ptr(skip), cfg.Cs, 0, 0, 0, 1,
                                    L.ptr(gW1), L.ptr(gb1), L.ptr(ws_for(cfg.Ce, cfg.Cs, P_f)), L.stream()),
                     reads=(da1,))
        dskip = _e(P_f, cfg.Cs, dev)
        skip_w_done = False
        gWs_all = None
        dg_skip = None        # throughput mode: [P_f][32*L] data gradients of all layers' skip convs, one product
        if ctx.W1_bf is not None:
            # dskip = (da1 @ W1) gated by skip > 0: W1 (Ce, Cs) row-major is the [K][N] operand as it lies
            fuse_skip = cfg.Cs % 32 == 0 and (32 * cfg.L) % 8 == 0
            dskip_bf = torch.empty((P_f, cfg.Cs), device=dev, dtype=torch.bfloat16) if fuse_skip else None
            L.call('mo_gemm_bf16_ex', L.ptr(da1_bf), cfg.Ce, L.ptr(ctx.W1_bf), cfg.Cs, 1, L.ptr(dskip), cfg.Cs,
                   P_f, cfg.Cs, cfg.Ce, 0, L.ptr(dskip_bf), None, 0, L.ptr(skip), st)
            if fuse_skip:
                # dg_i[crop] += dskip @ Ws_i for every layer i at once: Ws_i (Cs, 32) side by side is the [K][N] operand
                Wcat = torch.cat([p[f'skip_convs.{i}.weight'].reshape(cfg.Cs, 32) for i in range(cfg.L)], dim=1)
                Wcat_bf = torch.empty((cfg.Cs, 32 * cfg.L), device=dev, dtype=torch.bfloat16)
                L.call('mo_f32_to_bf16', L.ptr(Wcat), L.ptr(Wcat_bf), Wcat.numel(), st)
                dg_skip = _e(P_f, 32 * cfg.L, dev)
                L.call('mo_gemm_bf16', L.ptr(dskip_bf), cfg.Cs, L.ptr(Wcat_bf), 32 * cfg.L, 1,
                       L.ptr(dg_skip), 32 * cfg.L, P_f, 32 * cfg.L, cfg.Cs, 0, None, st)
                if P_f % 32 == 0 and cfg.Cs % 8 == 0:
                    # ... and their weight gradients one k-major product over the gathered crops of every layer's g
                    gWs_all = [gbuf(f'skip_convs.{i}.weight', p[f'skip_convs.{i}.weight']) for i in range(cfg.L)]

                    def _skip_w(dskip_bf=dskip_bf, gWs_all=gWs_all):
                        gcat = torch.empty((P_f, 32 * cfg.L), device=dev, dtype=torch.bfloat16)
                        touts = (_C.c_int * cfg.L)(*[ctx.layers[i]['g_T'] for i in range(cfg.L)])
                        L.call('mo_skip_gather_bf16', L.ptr_array([ctx.layers[i]['g'] for i in range(cfg.L)]), touts, cfg.L,
                               G, Tf, L.ptr(gcat), L.stream())
                        dW_all = _e(cfg.Cs, 32 * cfg.L, dev)
                        wsk = torch.empty(lib.mo_wgrad_bf16_kk_ws_floats(cfg.Cs, 32 * cfg.L, P_f), device=dev,
                                          dtype=torch.float32)
                        L.call('mo_wgrad_bf16_kk', L.ptr(dskip_bf), cfg.Cs, L.ptr(gcat), 32 * cfg.L, P_f, cfg.Cs, 32 * cfg.L,
                               L.ptr(dW_all), L.ptr(wsk), L.stream())
                        L.call('mo_skip_wsplit', L.ptr(dW_all), cfg.Cs, cfg.L, L.ptr_array(gWs_all), L.stream())
                    lane.run(_skip_w, reads=(dskip_bf,))
                    skip_w_done = True
        else:
            L.call('mo_conv1x1_bwd_data', L.ptr(da1), cfg.Ce, P_f, L.ptr(p['end_conv_1.weight']), cfg.Cs,
                   L.ptr(dskip), 0, 0, 0, L.ptr(skip), 0, st)
        grads['end_conv_2.weight'], grads['end_conv_2.bias'] = gW2, gb2
        grads['end_conv_1.weight'], grads['end_conv_1.bias'] = gW1, gb1
        dbs = torch.empty(cfg.Cs, device=dev, dtype=torch.float32)
        L.call('mo_colsum', L.ptr(dskip), P_f, cfg.Cs, L.ptr(dbs), L.ptr(ws), st)

        dAdp = None
        if cfg.gcn and cfg.adaptive:
            dAdp = _e(N, N, dev)
        dAdp_started = False
        dxo = None            # gradient w.r.t. the BatchNorm output of the current layer
        for i in reversed(range(cfg.L)):
            ly = ctx.layers[i]
            Tin, Tout = ly['Tin'], ly['Tout']
            P = G * Tout
            J = B * Tout * 32
            g = ly['g']
            srcs = ly['srcs']
            ns = len(srcs)
            dh = None
            if dxo is not None:
                # BatchNorm backward (graph_wavenet.py:250)
                dh = _e(P, 32, dev)
                gg = gbuf(f'bn.{i}.weight', shape=(32,)); gb = gbuf(f'bn.{i}.bias', shape=(32,))
                wsb = torch.empty(lib.mo_mlp_partial_floats(P) + 64, device=dev, dtype=torch.float32)
                L.call('mo_bn_bwd', L.ptr(dxo), L.ptr(ly['h']), P, L.ptr(p[f'bn.{i}.weight']),
                       L.ptr(ly['stats'][2]), L.ptr(ly['stats'][3]), L.ptr(dh), L.ptr(gg), L.ptr(gb),
                       L.ptr(wsb), st)
                grads[f'bn.{i}.weight'], grads[f'bn.{i}.bias'] = gg, gb
                # mlp / residual-conv backward (graph_wavenet.py:95-97 / :245)
                # (the gradient of source 0, the gated output, is the fp32 accumulator dg even where g itself is read as bf16)
                dsrcs = [torch.empty((P, 32), device=dev, dtype=torch.float32 if k_ == 0 else sr.dtype)
                         for k_, sr in enumerate(srcs)]
                smask = _bf_mask(srcs)
                dmask = smask & ~1
                if cfg.gcn:
                    W = p[f'gconv.{i}.mlp.mlp.weight']
                    kW, kb = f'gconv.{i}.mlp.mlp.weight', f'gconv.{i}.mlp.mlp.bias'
                else:
                    W = p[f'residual_convs.{i}.weight']
                    kW, kb = f'residual_convs.{i}.weight', f'residual_convs.{i}.bias'
                gW = gbuf(kW, W); gbm = gbuf(kb, shape=(32,))
                L.call('mo_gcn_mlp_bwd', L.ptr(dh), L.ptr_array(srcs), L.ptr_array(dsrcs), ns, L.ptr(W), P,
                       ly['seed'], ly['thresh'], ly['dscale'], L.ptr(gW), L.ptr(gbm), None, None, 1, smask, dmask,
                       st)

                def _mlp_w(dh=dh, srcs=srcs, dsrcs=dsrcs, ns=ns, W=W, P=P, ly=ly, gW=gW, gbm=gbm, smask=smask, dmask=dmask):
                    L.call('mo_gcn_mlp_bwd', L.ptr(dh), L.ptr_array(srcs), L.ptr_array(dsrcs), ns, L.ptr(W), P,
                           ly['seed'], ly['thresh'], ly['dscale'], L.ptr(gW), L.ptr(gbm),
                           L.ptr(ws_for(32, 32 * ns, P)), None, 2, smask, dmask, L.stream())
                lane.run(_mlp_w, reads=(dh,))
                grads[kW], grads[kb] = gW, gbm
                dg = dsrcs[0]
                k = 1
                if cfg.gcn:
                    ka = 1 + 2 * len(statics)
                    use_gs = bool(cfg.adaptive and statics and cfg.overlap and not SERIAL and OVL_DA)   # dA on its own stream
                    side = _side_stream(dev) if (use_gs and OVL_DENSE) else None
                    if cfg.adaptive:
                        # dense branch, part 1 (does not touch dg): dx1 += adp.dx2 ; dA += x1.dx2^T ; dA += g.dx1^T
                        x1 = srcs[ka]
                        dx1, dx2 = dsrcs[ka], dsrcs[ka + 1]
                        main = torch.cuda.current_stream()
                        if side is not None:
                            side.wait_stream(main)
                        with torch.cuda.stream(side if side is not None else main):
                            bf = ly['bf'] is not None
                            if bf:
                                g_bf, x1_bf = ly['bf']
                                _adj_prod(ctx.adp_bf, dx2, None, N, J, 1, dx1)        # dx1, dx2 are bf16 tensors
                            else:
                                _dense('mo_adj_gemm', N, J, L.ptr(ctx.adpT), N, L.ptr(dx2), L.ptr(dx1), J, 1, L.stream())

                            def _dA(beta0):
                                if bf:
                                    _adj_grad_bf(x1_bf, dx2, dAdp, N, J, beta0)
                                    _adj_grad_bf(g_bf, dx1, dAdp, N, J, 1)
                                else:
                                    _dense('mo_adj_grad', N, J, L.ptr(x1), L.ptr(dx2), N, J, L.ptr(dAdp), beta0, L.stream())
                                    _dense('mo_adj_grad', N, J, L.ptr(g), L.ptr(dx1), N, J, L.ptr(dAdp), 1, L.stream())
                            # only dx1 is needed by the main chain; the two dA accumulations keep running on a third
                            # stream beside the rest of this layer's backward
                            if use_gs:
                                dx1_ready = torch.cuda.Event()
                                dx1_ready.record(torch.cuda.current_stream())
                                gs = _side_stream((dev, 'dagrad'))
                                gs.wait_event(dx1_ready)
                                for t_ in (dx2, dx1):
                                    t_.record_stream(gs)
                                with torch.cuda.stream(gs):
                                    _dA(1 if dAdp_started else 0)
                            else:
                                _dA(1 if dAdp_started else 0)
                            dAdp_started = True
                    pend = []
                    for s in statics:
                        dx1s, dx2s = dsrcs[k], dsrcs[k + 1]
                        _spmm(s.bwd, N, dx2s, dx1s, J, 1)
                        pend.append((s.bwd, dx1s))
                        k += 2
                    # dg += S_a^T dx1_a + S_b^T dx1_b: the first hops of two supports in ONE pass over the fp32 dg
                    # (mo_spmm_blk2; same order of sums as the two launches)
                    while len(pend) >= 2 and _spmm2_ok(pend[0], pend[1], dg, J):
                        (ca, xa), (cb, xb) = pend[0], pend[1]
                        _spmm2(ca, xa, cb, xb, N, dg, J, 1)
                        pend = pend[2:]
                    for csr, x in pend:
                        _spmm(csr, N, x, dg, J, 1)
                    if cfg.adaptive:
                        if side is not None:
                            torch.cuda.current_stream().wait_event(dx1_ready)
                        # dense branch, part 2: dg += adp.dx1 (after the sparse accumulations into dg)
                        if ly['bf'] is not None:
                            _adj_prod(ctx.adp_bf, dx1, dg, N, J, 1)
                        else:
                            _dense('mo_adj_gemm', N, J, L.ptr(ctx.adpT), N, L.ptr(dx1), L.ptr(dg), J, 1, st)
                beta = 1
            else:
                dg = torch.zeros((P, 32), device=dev, dtype=torch.float32)
                beta = 1
            # skip path (graph_wavenet.py:230-236): dg[crop] += dskip @ Ws ; dWs
            Ws = p[f'skip_convs.{i}.weight']
            if dg_skip is not None:
                L.call('mo_skip_bwd_add', L.ptr(dg_skip), 32 * cfg.L, 32 * i, G, Tf, Tout, L.ptr(dg), st)
            else:
                L.call('mo_conv1x1_bwd_data', L.ptr(dskip), cfg.Cs, P_f, L.ptr(Ws), 32, L.ptr(dg), Tf, Tout,
                       Tout - Tf, None, beta, st)
            if skip_w_done:
                gWs = gWs_all[i]
            else:
                gWs = gbuf(f'skip_convs.{i}.weight', Ws)
                lane.run(lambda g=g, gWs=gWs, Tout=ly['g_T']: L.call(
                    'mo_conv1x1_bwd_weight', L.ptr(dskip), cfg.Cs, P_f, L.ptr(g), 32, Tf, Tout, Tout - Tf, 0,
                    L.ptr(gWs), None, L.ptr(ws_for(cfg.Cs, 32, P_f)), L.stream()), reads=(dskip,))
            grads[f'skip_convs.{i}.weight'] = gWs
            sb = gout.get(f'skip_convs.{i}.bias')
            if sb is not None:
                sb.copy_(dbs)
                grads[f'skip_convs.{i}.bias'] = sb
            else:
                grads[f'skip_convs.{i}.bias'] = dbs
            # gated TCN backward (graph_wavenet.py:222-226) + residual (:247)
            du = _e(G * Tin, 32, dev)
            gWf = gbuf(f'filter_convs.{i}.weight', p[f'filter_convs.{i}.weight']); gWg = gbuf(f'gate_convs.{i}.weight', p[f'gate_convs.{i}.weight'])
            gbf = gbuf(f'filter_convs.{i}.bias', shape=(32,)); gbg = gbuf(f'gate_convs.{i}.bias', shape=(32,))
            dpre = (torch.empty((P, 64), device=dev, dtype=torch.bfloat16) if ctx.mfma_bf16 else _e(P, 64, dev))
            ws2_dummy = torch.empty(16, device=dev, dtype=torch.float32)
            L.call('mo_tcn_bwd', L.ptr(ly['h_in']), L.ptr(ly['scale']), L.ptr(ly['shift']), L.ptr(ly['Wp']),
                   L.ptr(p[f'filter_convs.{i}.bias']), L.ptr(p[f'gate_convs.{i}.bias']), K, cfg.dil[i], G,
                   Tin, L.ptr(dg), L.ptr(dh), L.ptr(du), L.ptr(gWf), L.ptr(gWg), L.ptr(gbf), L.ptr(gbg),
                   L.ptr(dpre), L.ptr(ws2_dummy), 1, ctx.mfma_bf16, st)

            def _tcn_w(ly=ly, i=i, Tin=Tin, P=P, dg=dg, dpre=dpre, gWf=gWf, gWg=gWg, gbf=gbf, gbg=gbg):
                L.call('mo_tcn_bwd', L.ptr(ly['h_in']), L.ptr(ly['scale']), L.ptr(ly['shift']), L.ptr(ly['Wp']),
                       L.ptr(p[f'filter_convs.{i}.bias']), L.ptr(p[f'gate_convs.{i}.bias']), K, cfg.dil[i], G,
                       Tin, L.ptr(dg), None, None, L.ptr(gWf), L.ptr(gWg), L.ptr(gbf), L.ptr(gbg),
                       L.ptr(dpre), L.ptr(ws_for(64, 32 * K, P)), 2, ctx.mfma_bf16, L.stream())
            lane.run(_tcn_w, reads=(dpre, dg))
            grads[f'filter_convs.{i}.weight'], grads[f'filter_convs.{i}.bias'] = gWf, gbf
            grads[f'gate_convs.{i}.weight'], grads[f'gate_convs.{i}.bias'] = gWg, gbg
            if cfg.grad_ready is not None and i == cfg.L // 2 and cfg.L > 1:
                # data-parallel overlap: the head's and the late layers' gradients are final -- hand them to the
                # trainer's asynchronous all-reduce now, behind everything queued on the main and lane streams
                late = [n for n in cfg.names if n.startswith('end_conv_') or
                        (len(n.split('.')) > 2 and n.split('.')[1].isdigit() and int(n.split('.')[1]) >= i)]
                late = [n for n in late if n in gout and gout[n] is not None]
                if lane.side is not None:
                    lane.side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(lane.side):
                        cfg.grad_ready(late)
                else:
                    cfg.grad_ready(late)
            dxo = du

        # ---- start conv (graph_wavenet.py:191-196)
        pad = Tp - T
        Wst = p['start_conv.weight']
        gWst = gbuf('start_conv.weight', Wst); gbst = gbuf('start_conv.bias', shape=(32,))
        lane.run(lambda: L.call('mo_conv1x1_bwd_weight', L.ptr(dxo), 32, G * Tp, L.ptr(ctx.x_int), cfg.Cin, Tp, T, -pad,
                                0, L.ptr(gWst), L.ptr(gbst), L.ptr(ws_for(32, cfg.Cin, G * Tp)), L.stream()),
                 reads=(dxo,))
        grads['start_conv.weight'], grads['start_conv.bias'] = gWst, gbst
        dx = None
        if ctx.x_needs_grad:
            dx_int = _e(G * T, cfg.Cin, dev)
            L.call('mo_conv1x1_bwd_data', L.ptr(dxo), 32, G * Tp, L.ptr(Wst), cfg.Cin, L.ptr(dx_int), Tp, T,
                   -pad, None, 0, st)
            dx = torch.empty((B, cfg.Cin, N, T), device=dev, dtype=torch.float32)
            L.call('mo_nbtc_to_nchw', L.ptr(dx_int), L.ptr(dx), B, cfg.Cin, N, T, L.ptr(ctx.node_new), st)

        # ---- adaptive adjacency (graph_wavenet.py:202)
        if dAdp is not None and cfg.overlap and not SERIAL and statics:
            torch.cuda.current_stream().wait_stream(_side_stream((dev, 'dagrad')))   # the deferred dA accumulations
        if dAdp is not None:
            E1, E2 = p['nodevec1'], p['nodevec2']
            R = E1.shape[1]
            if dAdp_started:
                gE1 = gbuf('nodevec1', E1); gE2 = gbuf('nodevec2', E2)
                nz = (N + 127) // 128
                wsa = torch.empty(nz * R * N, device=dev, dtype=torch.float32)
                L.call('mo_adp_bwd', L.ptr(E1), L.ptr(E2), L.ptr(ctx.adp), L.ptr(dAdp), N, R, L.ptr(gE1),
                       L.ptr(gE2), L.ptr(wsa), wsa.numel(), st)
            else:
                gE1 = torch.zeros_like(E1); gE2 = torch.zeros_like(E2)
            grads['nodevec1'], grads['nodevec2'] = gE1, gE2

        lane.join()
        if cfg.grad_ready is not None:
            cfg.grad_ready([n for n in cfg.names if n in gout and gout[n] is not None])
        ctx.layers = ctx.x_int = ctx.adp = ctx.adpT = ctx.skip = ctx.r1 = ctx.adp_bf = ctx.W1_bf = ctx.skip_bf = None
        # gradients written straight into registered flat-buffer views are not handed to autograd
        return (None, None, None, None, None, dx) + tuple(
            None if (k in gout and gout[k] is not None) else grads[k] for k in cfg.names)


# ---------------------------------------------------------------------------------------------------------------------
# Small graphs, kernel_size 1, one window per call (the Graph WaveNet inside Modified_UNET, unet.py:221-226): the whole
# layer stack of a call is ONE workgroup (csrc/gwnet_small.hip), all calls of a step one launch; start conv, skip
# contraction, head and the boundary transposes are the general engine's kernels around it.
# ---------------------------------------------------------------------------------------------------------------------
def small_supported(cfg, n_static, T):
    lib = L.load()
    return bool(cfg.K == 1 and cfg.gcn and cfg.L <= 8 and cfg.rf == 1 and
                lib.mo_gwnet_small_supported(cfg.N, T, cfg.L, n_static + int(cfg.adaptive), n_static + int(cfg.adaptive)))


class GwnetSmallFunction(torch.autograd.Function):
    """y[b] = gwnet_body(x[b]) for every call b of a step, each call a batch of ONE window with its own BatchNorm
    statistics (graph_wavenet.py:191-254 as Modified_UNET calls it, unet.py:221-226).  x: (B, Cin, N, T) fp32.
    dense: the static supports as device tensors (N,N) or None for an identity support, in the reference's order."""

    @staticmethod
    def forward(ctx, cfg, dense, bn_bufs, training, x, *params):
        p = dict(zip(cfg.names, params))
        lib = L.load()
        dev = x.device
        st = L.stream()
        B, Cin, N, T = x.shape
        assert Cin == cfg.Cin and N == cfg.N and cfg.K == 1
        x = x.contiguous()
        G = N * B
        rows = G * T
        Lc = cfg.L
        x_int = _e(rows, Cin, dev)
        L.call('mo_nchw_to_nbtc', L.ptr(x), L.ptr(x_int), B, Cin, N, T, None, st)
        h0 = _e(rows, 32, dev)
        L.call('mo_conv1x1_fwd', L.ptr(x_int), Cin, 0, 0, 0, 0, L.ptr(p['start_conv.weight']), L.ptr(p['start_conv.bias']),
               32, L.ptr(h0), rows, 0, 0, st)
        adp = None
        mats = [d for d in dense if d is not None]
        if cfg.adaptive:
            adp = _e(N, N, dev)
            adpT = _e(N, N, dev)
            L.call('mo_adp_fwd', L.ptr(p['nodevec1']), L.ptr(p['nodevec2']), N, p['nodevec1'].shape[1], L.ptr(adp),
                   L.ptr(adpT), st)
            mats = mats + [adp]
        nsup = len(dense) + int(cfg.adaptive)
        dense_of, k = [], 0
        for d in list(dense) + ([adp] if cfg.adaptive else []):
            dense_of.append(-1 if d is None else k)
            k += d is not None
        nd = len(mats)
        c_dense = (_C.c_int * max(nsup, 1))(*dense_of)
        c_adj = L.ptr_array(mats) if mats else None
        plist = []
        for i in range(Lc):
            rm, rv = bn_bufs[i]
            plist += [p[f'filter_convs.{i}.weight'], p[f'filter_convs.{i}.bias'], p[f'gate_convs.{i}.weight'],
                      p[f'gate_convs.{i}.bias'], p[f'gconv.{i}.mlp.mlp.weight'], p[f'gconv.{i}.mlp.mlp.bias'],
                      p[f'bn.{i}.weight'], p[f'bn.{i}.bias'], rm, rv]
        c_params = L.ptr_array(plist)
        gcat = _e(rows, 32 * Lc, dev)
        hs = torch.empty((Lc, rows, 32), device=dev, dtype=torch.float32)
        xs = torch.empty((Lc, max(nd, 1), 2, rows, 32), device=dev, dtype=torch.float32) if nd else None
        stats = torch.empty((B, Lc, 6, 32), device=dev, dtype=torch.float32)
        drop_p = cfg.dropout if training else 0.0
        thresh = int(min(max(drop_p, 0.0), 0.999999) * 4294967296.0) if drop_p > 0 else 0
        dscale = 1.0 / (1.0 - drop_p) if drop_p > 0 else 1.0
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if thresh else 0
        L.call('mo_gwnet_small_fwd', B, N, T, Lc, nsup, c_dense, c_adj, c_params, L.ptr(h0), L.ptr(gcat), L.ptr(hs),
               L.ptr(xs), L.ptr(stats), 1 if training else 0, 1e-5, 0.1, seed, thresh, dscale, st)
        # skip path: all layers' skip convs are one contraction over the concatenated gated outputs (K = 32 L)
        Wcat = torch.cat([p[f'skip_convs.{i}.weight'].reshape(cfg.Cs, 32) for i in range(Lc)], dim=1).contiguous()
        bsum = torch.stack([p[f'skip_convs.{i}.bias'] for i in range(Lc)]).sum(0)
        skip = _e(rows, cfg.Cs, dev)
        L.call('mo_conv1x1_fwd', L.ptr(gcat), 32 * Lc, 0, 0, 0, 0, L.ptr(Wcat), L.ptr(bsum), cfg.Cs, L.ptr(skip), rows, 0, 0,
               st)
        r1 = _e(rows, cfg.Ce, dev)
        L.call('mo_conv1x1_fwd', L.ptr(skip), cfg.Cs, 0, 0, 0, 1, L.ptr(p['end_conv_1.weight']), L.ptr(p['end_conv_1.bias']),
               cfg.Ce, L.ptr(r1), rows, 1, 0, st)
        y_int = _e(rows, cfg.Cout, dev)
        L.call('mo_conv1x1_fwd', L.ptr(r1), cfg.Ce, 0, 0, 0, 0, L.ptr(p['end_conv_2.weight']), L.ptr(p['end_conv_2.bias']),
               cfg.Cout, L.ptr(y_int), rows, 0, 0, st)
        y = torch.empty((B, cfg.Cout, N, T), device=dev, dtype=torch.float32)
        L.call('mo_nbtc_to_nchw', L.ptr(y_int), L.ptr(y), B, cfg.Cout, N, T, None, st)
        ctx.cfg, ctx.training, ctx.params = cfg, training, p
        ctx.dims = (B, N, T, nsup, nd, dense_of)
        ctx.keep = (x_int, h0, adp, mats, plist, gcat, hs, xs, stats, skip, r1, Wcat)
        ctx.drop = (seed, thresh, dscale)
        ctx.x_needs_grad = x.requires_grad
        return y

    @staticmethod
    def backward(ctx, dy):
        cfg, p = ctx.cfg, ctx.params
        if not ctx.training:
            raise RuntimeError('gwnet backward is only defined in training mode (batch-stat BatchNorm)')
        lib = L.load()
        B, N, T, nsup, nd, dense_of = ctx.dims
        x_int, h0, adp, mats, plist, gcat, hs, xs, stats, skip, r1, Wcat = ctx.keep
        seed, thresh, dscale = ctx.drop
        dev = dy.device
        st = L.stream()
        Lc = cfg.L
        rows = N * B * T
        gout = cfg.grad_out or {}
        grads = {k: None for k in cfg.names}

        def gbuf(name, like=None, shape=None):
            t = gout.get(name)
            if t is not None:
                return t
            return torch.empty_like(like) if like is not None else torch.empty(shape, device=dev, dtype=torch.float32)

        def ws_for(M, Nn, P):
            return torch.empty(lib.mo_wgrad_ws_floats(M, Nn, P), device=dev, dtype=torch.float32)

        dy = dy.contiguous()
        dy_int = _e(rows, cfg.Cout, dev)
        L.call('mo_nchw_to_nbtc', L.ptr(dy), L.ptr(dy_int), B, cfg.Cout, N, T, None, st)
        # head (graph_wavenet.py:252-254)
        gW2 = gbuf('end_conv_2.weight', p['end_conv_2.weight']); gb2 = gbuf('end_conv_2.bias', p['end_conv_2.bias'])
        L.call('mo_conv1x1_bwd_weight', L.ptr(dy_int), cfg.Cout, rows, L.ptr(r1), cfg.Ce, 0, 0, 0, 0, L.ptr(gW2), L.ptr(gb2),
               L.ptr(ws_for(cfg.Cout, cfg.Ce, rows)), st)
        da1 = _e(rows, cfg.Ce, dev)
        L.call('mo_conv1x1_bwd_data', L.ptr(dy_int), cfg.Cout, rows, L.ptr(p['end_conv_2.weight']), cfg.Ce, L.ptr(da1),
               0, 0, 0, L.ptr(r1), 0, st)
        gW1 = gbuf('end_conv_1.weight', p['end_conv_1.weight']); gb1 = gbuf('end_conv_1.bias', p['end_conv_1.bias'])
        L.call('mo_conv1x1_bwd_weight', L.ptr(da1), cfg.Ce, rows, L.ptr(skip), cfg.Cs, 0, 0, 0, 1, L.ptr(gW1), L.ptr(gb1),
               L.ptr(ws_for(cfg.Ce, cfg.Cs, rows)), st)
        dskip = _e(rows, cfg.Cs, dev)
        L.call('mo_conv1x1_bwd_data', L.ptr(da1), cfg.Ce, rows, L.ptr(p['end_conv_1.weight']), cfg.Cs, L.ptr(dskip),
               0, 0, 0, L.ptr(skip), 0, st)
        grads['end_conv_2.weight'], grads['end_conv_2.bias'] = gW2, gb2
        grads['end_conv_1.weight'], grads['end_conv_1.bias'] = gW1, gb1
        # skip path: bias gradient (the same for every layer), data gradients of all layers in one product, weight
        # gradients of all layers in one product over the concatenated gated outputs
        dbs = torch.empty(cfg.Cs, device=dev, dtype=torch.float32)
        L.call('mo_colsum', L.ptr(dskip), rows, cfg.Cs, L.ptr(dbs),
               L.ptr(torch.empty(lib.mo_colsum_ws_floats(rows, cfg.Cs), device=dev, dtype=torch.float32)), st)
        dgskip = _e(rows, 32 * Lc, dev)
        L.call('mo_conv1x1_bwd_data', L.ptr(dskip), cfg.Cs, rows, L.ptr(Wcat), 32 * Lc, L.ptr(dgskip), 0, 0, 0, None, 0, st)
        dWcat = _e(cfg.Cs, 32 * Lc, dev)
        L.call('mo_conv1x1_bwd_weight', L.ptr(dskip), cfg.Cs, rows, L.ptr(gcat), 32 * Lc, 0, 0, 0, 0, L.ptr(dWcat), None,
               L.ptr(ws_for(cfg.Cs, 32 * Lc, rows)), st)
        gWs = [gbuf(f'skip_convs.{i}.weight', p[f'skip_convs.{i}.weight']) for i in range(Lc)]
        L.call('mo_skip_wsplit', L.ptr(dWcat), cfg.Cs, Lc, L.ptr_array(gWs), st)
        sbs = []
        for i in range(Lc):
            grads[f'skip_convs.{i}.weight'] = gWs[i]
            sb = gout.get(f'skip_convs.{i}.bias')
            if sb is not None:
                sbs.append(sb)
                grads[f'skip_convs.{i}.bias'] = sb
            else:
                grads[f'skip_convs.{i}.bias'] = dbs
        if sbs:
            torch._foreach_copy_(sbs, [dbs] * len(sbs))      # (one multi-tensor launch)
        # the layer stack
        dsts = []
        for i in range(Lc):
            last = i == Lc - 1               # the last layer's gcn / bn never reach the output (graph_wavenet.py:252)
            ds = [gbuf(f'filter_convs.{i}.weight', p[f'filter_convs.{i}.weight']), gbuf(f'filter_convs.{i}.bias', shape=(32,)),
                  gbuf(f'gate_convs.{i}.weight', p[f'gate_convs.{i}.weight']), gbuf(f'gate_convs.{i}.bias', shape=(32,))]
            if last:
                ds += [None, None, None, None]
            else:
                ds += [gbuf(f'gconv.{i}.mlp.mlp.weight', p[f'gconv.{i}.mlp.mlp.weight']),
                       gbuf(f'gconv.{i}.mlp.mlp.bias', shape=(32,)), gbuf(f'bn.{i}.weight', shape=(32,)),
                       gbuf(f'bn.{i}.bias', shape=(32,))]
            for key, t in zip((f'filter_convs.{i}.weight', f'filter_convs.{i}.bias', f'gate_convs.{i}.weight',
                               f'gate_convs.{i}.bias', f'gconv.{i}.mlp.mlp.weight', f'gconv.{i}.mlp.mlp.bias',
                               f'bn.{i}.weight', f'bn.{i}.bias'), ds):
                grads[key] = t
            dsts += ds
        c_dsts = (_C.c_void_p * len(dsts))(*[None if t is None else t.data_ptr() for t in dsts])
        c_dense = (_C.c_int * max(nsup, 1))(*dense_of)
        c_adj = L.ptr_array(mats) if mats else None
        c_params = L.ptr_array(plist)
        dh0 = _e(rows, 32, dev)
        ws = torch.empty(lib.mo_gwnet_small_bwd_ws_floats(B, N, T, Lc, nsup, nd), device=dev, dtype=torch.float32)
        dAdp = _e(N, N, dev) if cfg.adaptive else None
        L.call('mo_gwnet_small_bwd', B, N, T, Lc, nsup, c_dense, c_adj, (nd - 1) if cfg.adaptive else -1, c_params, c_dsts,
               L.ptr(h0), L.ptr(gcat), L.ptr(hs), L.ptr(xs), L.ptr(stats), 1e-5, seed, thresh, dscale, L.ptr(dgskip),
               L.ptr(dh0), L.ptr(ws), L.ptr(dAdp), st)
        # start conv (graph_wavenet.py:196)
        Wst = p['start_conv.weight']
        gWst = gbuf('start_conv.weight', Wst); gbst = gbuf('start_conv.bias', shape=(32,))
        L.call('mo_conv1x1_bwd_weight', L.ptr(dh0), 32, rows, L.ptr(x_int), cfg.Cin, 0, 0, 0, 0, L.ptr(gWst), L.ptr(gbst),
               L.ptr(ws_for(32, cfg.Cin, rows)), st)
        grads['start_conv.weight'], grads['start_conv.bias'] = gWst, gbst
        dx = None
        if ctx.x_needs_grad:
            dx_int = _e(rows, cfg.Cin, dev)
            L.call('mo_conv1x1_bwd_data', L.ptr(dh0), 32, rows, L.ptr(Wst), cfg.Cin, L.ptr(dx_int), 0, 0, 0, None, 0, st)
            dx = torch.empty((B, cfg.Cin, N, T), device=dev, dtype=torch.float32)
            L.call('mo_nbtc_to_nchw', L.ptr(dx_int), L.ptr(dx), B, cfg.Cin, N, T, None, st)
        if cfg.adaptive:
            E1, E2 = p['nodevec1'], p['nodevec2']
            R = E1.shape[1]
            gE1 = gbuf('nodevec1', E1); gE2 = gbuf('nodevec2', E2)
            if Lc > 1:
                nz = (N + 127) // 128
                wsa = torch.empty(nz * R * N, device=dev, dtype=torch.float32)
                L.call('mo_adp_bwd', L.ptr(E1), L.ptr(E2), L.ptr(adp), L.ptr(dAdp), N, R, L.ptr(gE1), L.ptr(gE2), L.ptr(wsa),
                       wsa.numel(), st)
            else:
                gE1.zero_(); gE2.zero_()
            grads['nodevec1'], grads['nodevec2'] = gE1, gE2
        if cfg.grad_ready is not None:
            cfg.grad_ready([n for n in cfg.names if n in gout and gout[n] is not None])
        ctx.keep = None                               # (release the saved activations with the backward pass)
        return (None, None, None, None, dx) + tuple(
            None if (k in gout and gout[k] is not None) else grads[k] for k in cfg.names)
