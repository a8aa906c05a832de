"""Graph WaveNet -- mirror of reference models/graph_wavenet.py (constructor, forward, state_dict).

Same class names, argument meaning and state_dict keys/shapes as the reference;
the computation of ``gwnet.forward`` (graph_wavenet.py:188-256) runs in the HIP engine
(multimodal_outage_amd.gwnet_engine).  Differences that are deliberate generalisations
(SURVEY.md 8b): module-level hyper-parameters became constructor kwargs with the reference values
as defaults; a 4-D (B,C,N,T) input bypasses the hard-wired views of :189/:255; a non-default
``supports`` argument is honoured (the reference silently replaces it with its module global, :120).
The nn.Conv2d / nn.BatchNorm2d sub-modules are parameter containers only (identical keys and
default initialisation); their forward() is never called.
"""
import numpy as np
import torch
import torch.nn as nn

from ..gwnet_engine import GwnetConfig, GwnetFunction, GwnetSmallFunction, StaticSupport, small_supported
from ._cache import tree_cache

# Hyperparameters (graph_wavenet.py:37-42)
image_dimension = 128
batch_size = 4
n_counties = 67
feature_vector_size = 256
loc_embed_size = 256
time_embed_size = 64

adjtype = "doubletransition"


def load_adj(filename, adjtype):
    """graph_wavenet.py:13-32: 'doubletransition' discards the CSV values and returns [I_N]."""
    if filename.endswith('.csv'):
        import pandas as pd
        adj_mx = pd.read_csv(filename, index_col=0).values
    else:
        raise NotImplementedError('pickle adjacency: the reference calls an undefined load_pickle '
                                  '(graph_wavenet.py:21)')
    if adjtype == "doubletransition":
        adj = [np.diag(np.ones(adj_mx.shape[0])).astype(np.float32)]
    else:
        error = 0
        assert error, "adj type not defined"
    return None, None, adj


class _DefaultSupports(list):
    """Marker for 'the reference's module-global default_supports' = [I_67] (graph_wavenet.py:50-51)."""


default_supports = _DefaultSupports([torch.eye(n_counties)])


class nconv(nn.Module):
    """graph_wavenet.py:60-66 (kept for API parity; gwnet.forward uses the fused engine)."""

    def forward(self, x, A):
        from .. import functional as MF
        return MF.nconv(x, A)


class linear(nn.Module):
    """graph_wavenet.py:68-74."""

    def __init__(self, c_in, c_out):
        super().__init__()
        self.mlp = torch.nn.Conv2d(c_in, c_out, kernel_size=(1, 1), padding=(0, 0), stride=(1, 1), bias=True)

    def forward(self, x):
        from .. import functional as MF
        return MF.conv1x1(x, self.mlp.weight, self.mlp.bias)


class gcn(nn.Module):
    """graph_wavenet.py:76-98 (parameter container + unfused forward for API parity)."""

    def __init__(self, c_in, c_out, dropout, support_len=3, order=2):
        super().__init__()
        self.nconv = nconv()
        c_in = (order * support_len + 1) * c_in
        self.mlp = linear(c_in, c_out)
        self.dropout = dropout
        self.order = order

    def forward(self, x, support):
        out = [x]
        for a in support:
            x1 = self.nconv(x, a)
            out.append(x1)
            for k in range(2, self.order + 1):
                x2 = self.nconv(x1, a)
                out.append(x2)
                x1 = x2
        h = torch.cat(out, dim=1)
        h = self.mlp(h)
        return torch.nn.functional.dropout(h, self.dropout, training=self.training)


class gwnet(nn.Module):
    def __init__(self, device, num_nodes=n_counties, dropout=0.3, supports=default_supports, gcn_bool=True,
                 addaptadj=True, aptinit=None, in_dim=feature_vector_size, out_dim=feature_vector_size - 1,
                 horizon=1, residual_channels=32, dilation_channels=32, skip_channels=256, end_channels=512,
                 kernel_size=1, blocks=4, layers=2):
        super(gwnet, self).__init__()
        if residual_channels != 32 or dilation_channels != 32:
            raise NotImplementedError('HIP path is specialised for residual=dilation=32 channels '
                                      '(the reference default, graph_wavenet.py:101)')
        self.dropout = dropout
        self.blocks = blocks
        self.layers = layers
        self.gcn_bool = gcn_bool
        self.addaptadj = addaptadj
        self.horizon = horizon
        self.num_nodes = num_nodes
        self.in_dim, self.out_dim, self.kernel_size = in_dim, out_dim, kernel_size
        self.skip_channels, self.end_channels = skip_channels, end_channels

        self.filter_convs = nn.ModuleList()
        self.gate_convs = nn.ModuleList()
        self.residual_convs = nn.ModuleList()
        self.skip_convs = nn.ModuleList()
        self.bn = nn.ModuleList()
        self.gconv = nn.ModuleList()

        self.start_conv = nn.Conv2d(in_channels=in_dim, out_channels=residual_channels, kernel_size=(1, 1))

        # graph_wavenet.py:120: self.supports = default_supports whatever the argument; a custom
        # argument is honoured here (generalisation), the default reproduces the reference ([I_N]).
        if supports is None:
            self.supports = None
        elif isinstance(supports, _DefaultSupports):
            self.supports = [np.eye(num_nodes, dtype=np.float32)]
        else:
            self.supports = [np.asarray(s.detach().cpu() if torch.is_tensor(s) else s, dtype=np.float32)
                             for s in supports]
            for s in self.supports:
                assert s.shape == (num_nodes, num_nodes)

        receptive_field = 1
        self.supports_len = 0
        if supports is not None:
            self.supports_len += len(supports)

        if gcn_bool and addaptadj:
            if self.supports is None:
                self.supports = []
            if aptinit is None:
                self.nodevec1 = nn.Parameter(torch.randn(num_nodes, 10), requires_grad=True)
                self.nodevec2 = nn.Parameter(torch.randn(10, num_nodes), requires_grad=True)
            else:
                m, p, n = torch.svd(aptinit)
                initemb1 = torch.mm(m[:, :10], torch.diag(p[:10] ** 0.5))
                initemb2 = torch.mm(torch.diag(p[:10] ** 0.5), n[:, :10].t())
                self.nodevec1 = nn.Parameter(initemb1, requires_grad=True)
                self.nodevec2 = nn.Parameter(initemb2, requires_grad=True)
            self.supports_len += 1

        for b in range(blocks):
            additional_scope = kernel_size - 1
            new_dilation = 1
            for i in range(layers):
                self.filter_convs.append(nn.Conv2d(residual_channels, dilation_channels,
                                                   kernel_size=(1, kernel_size), dilation=new_dilation))
                self.gate_convs.append(nn.Conv2d(residual_channels, dilation_channels,
                                                 kernel_size=(1, kernel_size), dilation=new_dilation))
                self.residual_convs.append(nn.Conv2d(dilation_channels, residual_channels, kernel_size=(1, 1)))
                self.skip_convs.append(nn.Conv2d(dilation_channels, skip_channels, kernel_size=(1, 1)))
                self.bn.append(nn.BatchNorm2d(residual_channels))
                new_dilation *= 2
                receptive_field += additional_scope
                additional_scope *= 2
                if self.gcn_bool:
                    self.gconv.append(gcn(dilation_channels, residual_channels, dropout,
                                          support_len=self.supports_len))

        self.end_conv_1 = nn.Conv2d(skip_channels, end_channels, kernel_size=(1, 1), bias=True)
        self.end_conv_2 = nn.Conv2d(end_channels, out_dim, kernel_size=(1, 1), bias=True)
        self.receptive_field = receptive_field
        self._statics = None
        self._statics_dev = None
        # 'f32': exact fp32 MFMA everywhere (parity mode); 'bf16': bf16 operands / fp32 accumulate for the
        # dense adaptive-adjacency products (throughput mode of BASELINE config 2)
        self.dense_dtype = 'f32'
        self.to(device)

    # ------------------------------------------------------------------ engine plumbing
    def _use_gcn(self):
        return self.gcn_bool and self.supports is not None

    def _engine_names(self):
        names = []
        use_gcn = self._use_gcn()
        if use_gcn and self.addaptadj:
            names += ['nodevec1', 'nodevec2']
        L = self.blocks * self.layers
        for i in range(L):
            names += [f'filter_convs.{i}.weight', f'filter_convs.{i}.bias',
                      f'gate_convs.{i}.weight', f'gate_convs.{i}.bias',
                      f'skip_convs.{i}.weight', f'skip_convs.{i}.bias',
                      f'bn.{i}.weight', f'bn.{i}.bias']
            if use_gcn:
                names += [f'gconv.{i}.mlp.mlp.weight', f'gconv.{i}.mlp.mlp.bias']
            else:
                names += [f'residual_convs.{i}.weight', f'residual_convs.{i}.bias']
        names += ['start_conv.weight', 'start_conv.bias', 'end_conv_1.weight', 'end_conv_1.bias',
                  'end_conv_2.weight', 'end_conv_2.bias']
        return names

    def _static_supports(self, device):
        """-> (supports, order, inverse): with static supports the nodes are renumbered by graph clusters so that the
        blocked SpMM of the throughput mode (mo_spmm_blk) can share neighbour rows; order[new] = old node.  Both
        numeric modes renumber, so the dropout mask (a function of seed and element index) is the same in both."""
        key = device
        if self._statics is None or self._statics_dev != key:
            sup = self.supports if self._use_gcn() else []
            if len(sup) * 2 + (2 if self.addaptadj else 0) + 1 > 11:
                raise NotImplementedError('HIP gcn mlp serves at most 5 supports (incl. adaptive): 11 column segments')
            order = None
            if len(sup) > 0:
                from ..gwnet_engine import cluster_order
                order = cluster_order([np.asarray(a) for a in sup])
            self._statics = [StaticSupport(a, device, order) for a in sup]
            if order is not None:
                inv = np.empty_like(order)
                inv[order] = np.arange(len(order))
                self._order = (torch.from_numpy(order).to(device), torch.from_numpy(inv.astype(np.int32)).to(device))
            else:
                self._order = (None, None)
            self._statics_dev = key
        return self._statics, self._order[0], self._order[1]

    def _apply(self, fn, *a, **kw):
        self.__dict__.pop('_mo_named', None)
        return super()._apply(fn, *a, **kw)

    def _config(self, names):
        cfg = GwnetConfig(num_nodes=self.num_nodes, in_dim=self.in_dim, out_dim=self.out_dim,
                          kernel_size=self.kernel_size, blocks=self.blocks, layers=self.layers,
                          skip_channels=self.skip_channels, end_channels=self.end_channels,
                          gcn=self._use_gcn(), adaptive=self._use_gcn() and self.addaptadj,
                          dropout=self.dropout, names=names)
        cfg.grad_out = getattr(self, '_mo_grad_out', None)
        cfg.grad_ready = getattr(self, '_mo_grad_ready', None)
        cfg.dense_bf16 = (getattr(self, 'dense_dtype', 'f32') == 'bf16')
        return cfg

    def body(self, x):
        """graph_wavenet.py:191-254 on a (B, in_dim, N, T) tensor -> (B, out_dim, N, T_final)."""
        if not x.is_cuda:
            raise RuntimeError('gwnet runs on the MI355X HIP path only (no CPU fallback); got a CPU tensor')
        names = self._engine_names()
        # (walked once; re-validated by object identity on every call, models/_cache.py)
        cache = tree_cache(self, '_mo_named')
        named = cache.named
        params = [named[k] for k in names]
        cfg = self._config(names)
        bn_bufs = [cache.bn[f'bn.{i}'][:2] for i in range(len(self.bn))]
        if self.training:
            for m in self.bn:
                m.num_batches_tracked += 1
        statics, order, inverse = self._static_supports(x.device)
        x = x.float()
        if order is not None:
            # the engine works in the renumbered node space: the input / output node axes are renumbered inside the
            # engine's boundary transposes (mo_nchw_to_nbtc / mo_nbtc_to_nchw take the map); only the (N,10) node
            # embeddings are gathered here (their gradients return through autograd)
            for k, name in enumerate(names):
                if name == 'nodevec1':
                    params[k] = params[k].index_select(0, order)
                elif name == 'nodevec2':
                    params[k] = params[k].index_select(1, order)
            if cfg.grad_out is not None:
                cfg.grad_out = {k: v for k, v in cfg.grad_out.items() if k not in ('nodevec1', 'nodevec2')}
        return GwnetFunction.apply(cfg, statics, bn_bufs, self.training, inverse, x, *params)

    def forward(self, input):
        if input.dim() == 3:
            # reference path: raw reinterpretation (view, not permute) of (N, H, C) -- :189 / :255
            x = input.contiguous().view(1, self.in_dim, self.num_nodes, self.horizon)
            y = self.body(x)
            return y.view(self.num_nodes, self.horizon, self.out_dim)
        return self.body(input)

    # ------------------------------------------------------------------ several reference-style calls in one launch
    def _small_dense(self, device):
        """The static supports for the small-graph kernel: None for an identity matrix (the reference's default support,
        graph_wavenet.py:13-32 -- folded into the mlp weights), else the dense (N,N) matrix on the device."""
        c = self.__dict__.get('_mo_small_dense')
        if c is None or c[0] != device:
            sup = self.supports if self._use_gcn() else []
            eye = np.eye(self.num_nodes, dtype=np.float32)
            mats = [None if np.array_equal(np.asarray(a, dtype=np.float32), eye)
                    else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device) for a in sup]
            c = self.__dict__['_mo_small_dense'] = (device, mats)
        return c[1]

    def forward_calls(self, inputs):
        """``torch.stack([self(inputs[b]) for b in range(B)])`` for reference-style 3-D inputs (B, N, H, in_dim) ->
        (B, N, H, out_dim): the per-batch-element calls of Modified_UNET.forward (unet.py:221-226), every call a batch of
        one window with its own BatchNorm statistics and its own sequential running-statistic update.  On a small graph
        with kernel_size 1 (the reference default) all calls run as ONE launch of the small-graph kernel
        (csrc/gwnet_small.hip: a call = one workgroup); anything else loops over the general engine."""
        B = inputs.shape[0]
        cfg = self._config(self._engine_names()) if inputs.is_cuda else None
        statics_n = len(self.supports) if (self._use_gcn() and self.supports) else 0
        if (cfg is None or inputs.dim() != 4 or not small_supported(cfg, statics_n, self.horizon)
                or getattr(self, 'small_graph_kernel', True) is False):
            if B > 1:
                self._mo_grad_out = None       # several backward passes through the engine must accumulate via autograd
            return torch.stack([self(inputs[b]) for b in range(B)])
        names = cfg.names
        cache = tree_cache(self, '_mo_named')
        params = [cache.named[k] for k in names]
        bn_bufs = [cache.bn[f'bn.{i}'][:2] for i in range(len(self.bn))]
        if self.training:
            # one BatchNorm2d call per layer and forward call (one multi-tensor launch instead of one per layer)
            torch._foreach_add_([m.num_batches_tracked for m in self.bn], B)
        x = inputs.float().contiguous().view(B, self.in_dim, self.num_nodes, self.horizon)    # :189, per call
        y = GwnetSmallFunction.apply(cfg, self._small_dense(inputs.device), bn_bufs, self.training, x, *params)
        return y.view(B, self.num_nodes, self.horizon, self.out_dim)                           # :255, per call
