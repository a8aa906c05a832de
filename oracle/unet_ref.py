"""Oracle: Modified_UNET forward, restated from reference models/unet.py (per-county Python
loops, per-call train-mode BatchNorm -- SURVEY.md F6/F7).  ``p`` = state_dict-keyed tensors."""
import torch
import torch.nn.functional as F

from .gwnet_ref import gwnet_forward


def _bn(p, key, x, training):
    if training:
        p[key + '.num_batches_tracked'].add_(1)
    return F.batch_norm(x, p[key + '.running_mean'], p[key + '.running_var'],
                        p[key + '.weight'], p[key + '.bias'], training, 0.1, 1e-5)


def double_conv(p, pre, x, training):
    """unet.py:40-53: [conv3x3 pad1 no-bias -> BN -> ReLU] x2."""
    x = F.conv2d(x, p[pre + '.double_conv.0.weight'], None, padding=1)
    x = F.relu(_bn(p, pre + '.double_conv.1', x, training))
    x = F.conv2d(x, p[pre + '.double_conv.3.weight'], None, padding=1)
    x = F.relu(_bn(p, pre + '.double_conv.4', x, training))
    return x


def down(p, pre, x, training):
    """unet.py:55-65."""
    return double_conv(p, pre + '.maxpool_conv.1', F.max_pool2d(x, 2), training)


def up(p, pre, x1, x2, training):
    """unet.py:67-84: convT(k2,s2) -> pad to skip size -> cat([skip, up]) -> DoubleConv."""
    x1 = F.conv_transpose2d(x1, p[pre + '.up.weight'], p[pre + '.up.bias'], stride=2)
    dX = x2.size(3) - x1.size(3)
    dY = x2.size(2) - x1.size(2)
    x1 = F.pad(x1, (dX // 2, dX - dX // 2, dY // 2, dY - dY // 2))
    return double_conv(p, pre + '.conv', torch.cat([x2, x1], dim=1), training)


def contraction(p, inp, horizon, training, pre='contraction'):
    """unet.py:106-126.  inp: (n_counties, H, Cin, h, w)."""
    n_counties = inp.shape[0]
    fmaps = [[] for _ in range(4)]
    enc = []
    for c in range(n_counties):
        x1 = double_conv(p, pre + '.inc', inp[c], training)
        x2 = down(p, pre + '.down1', x1, training)
        x3 = down(p, pre + '.down2', x2, training)
        x4 = down(p, pre + '.down3', x3, training)
        x5 = down(p, pre + '.down4', x4, training)
        for k, t in enumerate((x1, x2, x3, x4)):
            fmaps[k].append(t)
        enc.append(x5)
    fmaps = [torch.stack(f) for f in fmaps]
    enc = torch.stack(enc).view(n_counties, horizon, -1)
    return enc, fmaps


def fc_block(p, pre, x, drop_p, training):
    """unet.py:138-149 / :162-173: relu(fc1) -> dropout -> relu(fc2), per county."""
    out = []
    for c in range(x.shape[0]):
        h = torch.relu(F.linear(x[c], p[pre + '.fc1.weight'], p[pre + '.fc1.bias']))
        h = F.dropout(h, drop_p, training)
        h = torch.relu(F.linear(h, p[pre + '.fc2.weight'], p[pre + '.fc2.bias']))
        out.append(h)
    return torch.stack(out)


def expansion(p, x, fmaps, training, pre='expansion'):
    """unet.py:184-199."""
    preds = []
    for c in range(x.shape[0]):
        y = up(p, pre + '.up1', x[c], fmaps[-1][c], training)
        y = up(p, pre + '.up2', y, fmaps[-2][c], training)
        y = up(p, pre + '.up3', y, fmaps[-3][c], training)
        y = up(p, pre + '.up4', y, fmaps[-4][c], training)
        preds.append(F.conv2d(y, p[pre + '.outc.conv.weight'], p[pre + '.outc.conv.bias']))
    return torch.stack(preds)


def modified_unet_forward(p, inp, time_dim, *, horizon, supports, training=True, fc_dropout=0.0,
                          gw_dropout=0.0, gw_kwargs=None):
    """unet.py:219-231.  inp: (B, n_counties, H, Cin, h, w); time_dim: (B, n_counties, H, 64)."""
    gw_kwargs = dict(gw_kwargs or {})
    n_counties = inp.shape[1]
    gp = _SubDict(p, 'st_gnn.')
    res = []
    for b in range(inp.shape[0]):
        enc, fmaps = contraction(p, inp[b], horizon, training)
        feat = fc_block(p, 'encoder', enc, fc_dropout, training)
        feat = torch.cat((feat, time_dim[b]), dim=-1)
        in_dim = feat.shape[-1]
        x = feat.contiguous().view(1, in_dim, n_counties, horizon)          # graph_wavenet.py:189
        y = gwnet_forward(gp, x, supports=supports, dropout=gw_dropout, training=training,
                          **gw_kwargs)
        y = y.reshape(n_counties, horizon, -1)                               # graph_wavenet.py:255
        dec = fc_block(p, 'decoder', y, fc_dropout, training)
        s = int(round((dec.shape[-1] // 64) ** 0.5))
        dec = dec.view(n_counties, horizon, 64, s, s)
        res.append(expansion(p, dec, fmaps, training))
    return torch.stack(res)


class _SubDict:
    """View of a flat state dict under a key prefix (``st_gnn.``)."""

    def __init__(self, d, prefix):
        self.d, self.prefix = d, prefix

    def __getitem__(self, k):
        return self.d[self.prefix + k]

    def __setitem__(self, k, v):
        self.d[self.prefix + k] = v


def date2vec_encode(x, fc1_w, fc1_b, fc2_w, fc2_b):
    """date2vec.py:49-53 (act='sin'): cat[fc1(x), sin(fc2(x))] on dim 1."""
    return torch.cat([F.linear(x, fc1_w, fc1_b), torch.sin(F.linear(x, fc2_w, fc2_b))], 1)
