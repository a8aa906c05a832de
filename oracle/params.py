"""Oracle: state_dict schemas (ordered key -> shape) of the reference modules, and the
deterministic seeded fill shared by the golden generator and the tests.

Schema follows SURVEY.md App. B / graph_wavenet.py:101-185 / unet.py:40-217: direct parameters
first (nodevec1/2), then sub-modules in attribute-creation order.
"""
from collections import OrderedDict

import numpy as np
import torch


def _conv(d, pre, co, ci, kh, kw, bias=True):
    d[pre + '.weight'] = (co, ci, kh, kw)
    if bias:
        d[pre + '.bias'] = (co,)


def _bn(d, pre, c):
    d[pre + '.weight'] = (c,)
    d[pre + '.bias'] = (c,)
    d[pre + '.running_mean'] = (c,)
    d[pre + '.running_var'] = (c,)
    d[pre + '.num_batches_tracked'] = ()


def gwnet_schema(num_nodes=67, supports_len=2, in_dim=256, out_dim=255, residual_channels=32,
                 dilation_channels=32, skip_channels=256, end_channels=512, kernel_size=1,
                 blocks=4, layers=2, gcn_bool=True, addaptadj=True, prefix=''):
    """supports_len counts static supports + 1 adaptive (graph_wavenet.py:124-134)."""
    d = OrderedDict()
    L = blocks * layers
    if gcn_bool and addaptadj:
        d['nodevec1'] = (num_nodes, 10)
        d['nodevec2'] = (10, num_nodes)
    for i in range(L):
        _conv(d, f'filter_convs.{i}', dilation_channels, residual_channels, 1, kernel_size)
    for i in range(L):
        _conv(d, f'gate_convs.{i}', dilation_channels, residual_channels, 1, kernel_size)
    for i in range(L):
        _conv(d, f'residual_convs.{i}', residual_channels, dilation_channels, 1, 1)
    for i in range(L):
        _conv(d, f'skip_convs.{i}', skip_channels, dilation_channels, 1, 1)
    for i in range(L):
        _bn(d, f'bn.{i}', residual_channels)
    if gcn_bool:
        for i in range(L):
            _conv(d, f'gconv.{i}.mlp.mlp', residual_channels,
                  (2 * supports_len + 1) * dilation_channels, 1, 1)
    _conv(d, 'start_conv', residual_channels, in_dim, 1, 1)
    _conv(d, 'end_conv_1', end_channels, skip_channels, 1, 1)
    _conv(d, 'end_conv_2', out_dim, end_channels, 1, 1)
    return OrderedDict((prefix + k, v) for k, v in d.items())


def _double_conv(d, pre, ci, co):
    d[pre + '.double_conv.0.weight'] = (co, ci, 3, 3)
    _bn(d, pre + '.double_conv.1', co)
    d[pre + '.double_conv.3.weight'] = (co, co, 3, 3)
    _bn(d, pre + '.double_conv.4', co)


def unet_schema(input_channels=1, output_channels=1, image_dimension=128, n_counties=67,
                feature_vector_size=256, time_embed_size=64, compression_factor=4,
                gwnet_kwargs=None):
    """Modified_UNET(st_gnn='gwnet') (unet.py:202-217)."""
    d = OrderedDict()
    _double_conv(d, 'contraction.inc', input_channels, 4)
    for k, (ci, co) in enumerate(((4, 8), (8, 16), (16, 32), (32, 64)), 1):
        _double_conv(d, f'contraction.down{k}.maxpool_conv.1', ci, co)
    first = int((image_dimension / 16) ** 2 * 64)
    d['encoder.fc1.weight'] = (first // compression_factor, first)
    d['encoder.fc1.bias'] = (first // compression_factor,)
    d['encoder.fc2.weight'] = (feature_vector_size, first // compression_factor)
    d['encoder.fc2.bias'] = (feature_vector_size,)
    gk = dict(num_nodes=n_counties, supports_len=2,
              in_dim=feature_vector_size + time_embed_size, out_dim=feature_vector_size)
    gk.update(gwnet_kwargs or {})
    d.update(gwnet_schema(prefix='st_gnn.', **gk))
    d['decoder.fc1.weight'] = (feature_vector_size * compression_factor, feature_vector_size)
    d['decoder.fc1.bias'] = (feature_vector_size * compression_factor,)
    d['decoder.fc2.weight'] = (first, feature_vector_size * compression_factor)
    d['decoder.fc2.bias'] = (first,)
    for k, (ci, co) in enumerate(((64, 32), (32, 16), (16, 8), (8, 4)), 1):
        d[f'expansion.up{k}.up.weight'] = (ci, ci // 2, 2, 2)
        d[f'expansion.up{k}.up.bias'] = (ci // 2,)
        _double_conv(d, f'expansion.up{k}.conv', ci, co)
    d['expansion.outc.conv.weight'] = (output_channels, 4, 1, 1)
    d['expansion.outc.conv.bias'] = (output_channels,)
    return d


def seeded_values(schema, seed):
    """Deterministic values for every entry of ``schema`` (numpy RandomState(seed), consumed in
    key order).  Weights ~ U(+-1.7/sqrt(fan_in)), BN gamma U(.5,1.5), biases U(+-.3),
    running_var U(.5,1.5), running_mean U(+-.2), nodevecs N(0,1), num_batches_tracked 0."""
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for name, shape in schema.items():
        if name.endswith('num_batches_tracked'):
            out[name] = torch.zeros((), dtype=torch.long)
            continue
        if name.endswith('running_var'):
            v = rs.uniform(0.5, 1.5, size=shape)
        elif name.endswith('running_mean'):
            v = rs.uniform(-0.2, 0.2, size=shape)
        elif 'nodevec' in name:
            v = rs.standard_normal(size=shape)
        elif len(shape) <= 1:
            if name.endswith('weight'):
                v = rs.uniform(0.5, 1.5, size=shape)
            else:
                v = rs.uniform(-0.3, 0.3, size=shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            v = rs.uniform(-1.0, 1.0, size=shape) * (1.7 / np.sqrt(fan_in))
        out[name] = torch.from_numpy(np.asarray(v, dtype=np.float32)).clone()
    return out


def load_into(module, values):
    """Copy seeded values into a torch module whose state_dict has the same keys/shapes."""
    sd = module.state_dict()
    assert list(sd.keys()) == list(values.keys()), \
        [k for k in sd.keys() if k not in values] + [k for k in values if k not in sd]
    with torch.no_grad():
        for k, t in sd.items():
            assert tuple(t.shape) == tuple(values[k].shape), (k, t.shape, values[k].shape)
            t.copy_(values[k])


def as_param_dict(values, requires_grad=True):
    """Clone seeded values into a dict of leaf tensors for the functional oracle."""
    out = {}
    for k, v in values.items():
        t = v.clone()
        if requires_grad and t.is_floating_point() and not (
                k.endswith('running_mean') or k.endswith('running_var')):
            t.requires_grad_(True)
        out[k] = t
    return out


def knn_graph(n, mean_degree=6, seed=0):
    """Synthetic adjacency of SURVEY.md 8(d): symmetric k-NN graph of n uniform points in the
    unit square (RandomState(seed)), 0/1 values, zero diagonal, mean degree ~ mean_degree."""
    rs = np.random.RandomState(seed)
    pts = rs.uniform(size=(n, 2))
    k = max(1, int(round(mean_degree * 0.82)))
    A = np.zeros((n, n), dtype=np.float32)
    # blockwise distance computation to stay small in memory
    for s in range(0, n, 512):
        d = ((pts[s:s + 512, None, :] - pts[None, :, :]) ** 2).sum(-1)
        d[np.arange(d.shape[0]), np.arange(s, s + d.shape[0])] = np.inf
        nb = np.argpartition(d, k, axis=1)[:, :k]
        for r in range(d.shape[0]):
            A[s + r, nb[r]] = 1.0
    A = np.maximum(A, A.T)
    np.fill_diagonal(A, 0.0)
    return A
