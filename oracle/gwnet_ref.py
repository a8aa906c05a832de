"""Oracle: Graph WaveNet forward, restated from reference models/graph_wavenet.py.

Functional, reference-faithful (dense einsum supports, unfused ops).  ``p`` is a dict holding the
reference's state_dict tensors (keys as SURVEY.md App. B, without prefix); differentiable
w.r.t. every tensor in ``p`` that has requires_grad.
"""
import torch
import torch.nn.functional as F


def receptive_field(kernel_size, blocks, layers):
    """graph_wavenet.py:122,146-170."""
    rf = 1
    for _ in range(blocks):
        scope = kernel_size - 1
        for _ in range(layers):
            rf += scope
            scope *= 2
    return rf


def dilations(blocks, layers):
    """graph_wavenet.py:145-168 (new_dilation doubles per layer, resets per block)."""
    return [2 ** i for _ in range(blocks) for i in range(layers)]


def nconv(x, A):
    """graph_wavenet.py:64-66: out[n,c,w,l] = sum_v x[n,c,v,l] * A[v,w]."""
    return torch.einsum('ncvl,vw->ncwl', (x, A)).contiguous()


def adaptive_adj(nodevec1, nodevec2):
    """graph_wavenet.py:202."""
    return F.softmax(F.relu(torch.mm(nodevec1, nodevec2)), dim=1)


def gcn(x, supports, w, b, dropout, training, order=2):
    """graph_wavenet.py:85-98."""
    out = [x]
    for a in supports:
        x1 = nconv(x, a)
        out.append(x1)
        for _ in range(2, order + 1):
            x2 = nconv(x1, a)
            out.append(x2)
            x1 = x2
    h = torch.cat(out, dim=1)
    h = F.conv2d(h, w, b)
    h = F.dropout(h, dropout, training=training)
    return h


def batchnorm_train(x, weight, bias, running_mean, running_var, momentum=0.1, eps=1e-5,
                    training=True):
    """nn.BatchNorm2d semantics (graph_wavenet.py:167,250); updates running stats in place."""
    return F.batch_norm(x, running_mean, running_var, weight, bias, training, momentum, eps)


def gwnet_forward(p, x, *, supports, blocks=4, layers=2, kernel_size=1, dropout=0.0,
                  training=True, gcn_bool=True, addaptadj=True, return_intermediates=False):
    """graph_wavenet.py:191-254 (the shape-generic body; the :189/:255 views are the caller's).

    x: (B, Cin, N, T).  supports: list of dense (N,N) tensors or None.  BN running stats in ``p``
    (``bn.i.running_mean`` ...) are updated in place when training, as nn.BatchNorm2d does.
    """
    rf = receptive_field(kernel_size, blocks, layers)
    dil = dilations(blocks, layers)
    inter = {}
    in_len = x.size(3)
    if in_len < rf:
        x = F.pad(x, (rf - in_len, 0, 0, 0))
    x = F.conv2d(x, p['start_conv.weight'], p['start_conv.bias'])
    skip = 0
    new_supports = None
    if gcn_bool and addaptadj and supports is not None:
        adp = adaptive_adj(p['nodevec1'], p['nodevec2'])
        new_supports = list(supports) + [adp]
        inter['adp'] = adp
    for i in range(blocks * layers):
        residual = x
        filt = torch.tanh(F.conv2d(residual, p[f'filter_convs.{i}.weight'],
                                   p[f'filter_convs.{i}.bias'], dilation=(1, dil[i])))
        gate = torch.sigmoid(F.conv2d(residual, p[f'gate_convs.{i}.weight'],
                                      p[f'gate_convs.{i}.bias'], dilation=(1, dil[i])))
        x = filt * gate
        inter[f'gated.{i}'] = x
        s = F.conv2d(x, p[f'skip_convs.{i}.weight'], p[f'skip_convs.{i}.bias'])
        if isinstance(skip, int):
            skip = 0
        else:
            skip = skip[:, :, :, -s.size(3):]
        skip = s + skip
        if gcn_bool and supports is not None:
            sup = new_supports if addaptadj else supports
            x = gcn(x, sup, p[f'gconv.{i}.mlp.mlp.weight'], p[f'gconv.{i}.mlp.mlp.bias'],
                    dropout, training)
        else:
            x = F.conv2d(x, p[f'residual_convs.{i}.weight'], p[f'residual_convs.{i}.bias'])
        x = x + residual[:, :, :, -x.size(3):]
        inter[f'prebn.{i}'] = x
        if training:
            p[f'bn.{i}.num_batches_tracked'].add_(1)
        x = batchnorm_train(x, p[f'bn.{i}.weight'], p[f'bn.{i}.bias'],
                            p[f'bn.{i}.running_mean'], p[f'bn.{i}.running_var'],
                            training=training)
    inter['skip'] = skip
    x = F.relu(skip)
    x = F.relu(F.conv2d(x, p['end_conv_1.weight'], p['end_conv_1.bias']))
    x = F.conv2d(x, p['end_conv_2.weight'], p['end_conv_2.bias'])
    if return_intermediates:
        return x, inter
    return x


def gwnet_forward_ref_views(p, x3, *, horizon, supports, n_nodes=67, in_dim=320, out_dim=256,
                            **kw):
    """graph_wavenet.py:189 and :255: raw memory reinterpretation (view, not permute) of the
    (N, H, Cin) tensor to (1, Cin, N, H), and of the (1, out, N, H) result to (N, H, out)."""
    x = x3.view(1, in_dim, n_nodes, horizon)
    y = gwnet_forward(p, x, supports=supports, **kw)
    return y.reshape(n_nodes, horizon, out_dim)


def asym_adj(adj):
    """utils.py:152-158 restated in numpy: D^-1 A with inf -> 0, float32 result.  The dtype flow
    of the reference is kept (rowsum and the reciprocal in the input dtype) so 0/1 float32
    adjacencies give bit-identical values."""
    import numpy as np
    adj = np.asarray(adj)
    rowsum = np.asarray(adj.sum(1)).flatten()
    with np.errstate(divide='ignore'):
        d_inv = np.power(rowsum, -1).flatten()
    d_inv[np.isinf(d_inv)] = 0.
    return (d_inv[:, None] * adj).astype(np.float32)
