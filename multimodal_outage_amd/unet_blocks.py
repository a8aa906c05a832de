"""Callable UNet sub-blocks on the HIP engine (reference models/unet.py:40-199).

``Modified_UNET.forward`` runs the whole encoder / decoder as two fused autograd Functions (unet_engine.py); the
reference's building blocks -- DoubleConv, Down, Up, OutConv, Contraction, Encoder, Decoder, Expansion -- are part of
its public surface too (``forward`` signatures kept verbatim), so each of them is callable on its own through the same
engine pieces: one autograd Function per block, inputs and outputs are ordinary (activated) tensors.  Semantics follow
the reference call by call: a standalone DoubleConv / Down / Up normalises over the batch it is given (one BatchNorm
group, one running-stat update); Contraction / Expansion run one group per county (``horizon`` images) with the
reference's county-order sequence of running-stat updates.
"""
import torch
import torch.nn as nn

from . import _lib as L
from .unet_engine import (View, Grads, _empty, _NOVIEW, double_conv_fwd, double_conv_bwd, fc_block_fwd, fc_block_bwd,
                          _drop_params, ENC_CH, DEC_CH)


def _need_gpu(t):
    if not t.is_cuda:
        raise RuntimeError('the UNet blocks run on the MI355X HIP path only (no CPU fallback)')


def _materialise(v, n, gs, pool=0):
    """Activated copy of a view (optionally 2x2 max-pooled)."""
    Ho, Wo = (v.H // 2, v.W // 2) if pool else (v.H, v.W)
    out = _empty(n, v.C, Ho, Wo, dev=v.t.device)
    L.call('mo_unet_act', L.ptr(v.t), v.istride, v.C, n, v.H, v.W, L.ptr(v.sc), L.ptr(v.sh), gs, pool, L.ptr(out),
           v.C * Ho * Wo, L.BF_IN0 * v.bf, L.stream())
    return out


class _Cfg:
    """Non-tensor arguments of a block Function."""

    def __init__(self, names, bufs, training, **kw):
        self.names, self.bufs, self.training = names, bufs, training
        self.__dict__.update(kw)


class DoubleConvFn(torch.autograd.Function):
    """unet.py:40-53 (+ the MaxPool2d of Down, :55-65, when cfg.pool): y = relu(bn(conv(relu(bn(conv(pool?(x)))))))."""

    @staticmethod
    def forward(ctx, cfg, x, *params):
        p = dict(zip(cfg.names, params))
        x = x.contiguous().float()
        n, Ci, H, W = x.shape
        dev = x.device
        gs = cfg.gsize or n
        xin = x
        if cfg.pool:
            xin = _materialise(View(x, Ci, H, W), n, gs, pool=1)
            H, W = H // 2, W // 2
        sv, v = double_conv_fwd(p, cfg.pre, [View(xin, Ci, H, W)], cfg.Co, n, gs, cfg.training, cfg.bufs, dev)
        out = _materialise(v, n, gs)
        ctx.cfg, ctx.p, ctx.sv, ctx.n, ctx.gs = cfg, p, sv, n, gs
        ctx.x = x if cfg.pool else None
        ctx.need_dx = x.requires_grad
        return out

    @staticmethod
    def backward(ctx, dout):
        cfg, p, n = ctx.cfg, ctx.p, ctx.n
        dev = dout.device
        grads = Grads(None, dev)
        dx = double_conv_bwd(p, ctx.sv, n, ctx.gs, grads, dev, da=dout.contiguous(), dp=None, need_input_grad=ctx.need_dx)
        if dx is not None and cfg.pool:
            x = ctx.x
            _, Ci, H, W = x.shape
            full = _empty(n, Ci, H, W, dev=dev)
            L.call('mo_maxpool2_bwd', L.ptr(x), Ci * H * W, Ci, n, H, W, L.ptr(dx), Ci * (H // 2) * (W // 2), L.ptr(full),
                   Ci * H * W, L.stream())
            dx = full
        return (None, dx) + grads.result(cfg.names)


class UpFn(torch.autograd.Function):
    """unet.py:67-84: up = ConvTranspose2d(x1); cat([x2, up]); DoubleConv."""

    @staticmethod
    def forward(ctx, cfg, x1, x2, *params):
        p = dict(zip(cfg.names, params))
        x1, x2 = x1.contiguous().float(), x2.contiguous().float()
        n, Ci, H, W = x1.shape
        dev = x1.device
        gs = cfg.gsize or n
        C0 = Ci // 2
        H2, W2 = x2.shape[2], x2.shape[3]
        u = _empty(n, C0, 2 * H, 2 * W, dev=dev)
        L.call('mo_convt2x2_fwd', L.ptr(x1), Ci * H * W, Ci, None, None, 0, gs, L.ptr(p[cfg.pre + '.up.weight']),
               L.ptr(p[cfg.pre + '.up.bias']), C0, n, H, W, L.ptr(u), C0 * 4 * H * W, 0, L.stream())
        pad = None
        if (H2, W2) != (2 * H, 2 * W):
            # unet.py:76-81: the upsampled map is zero-padded (or cropped: F.pad takes negative widths) to the skip map's
            # size.  Never reached inside Modified_UNET for image sizes that are multiples of 16; here it is a
            # materialised copy of the small upsampled map (torch), the convs then run on the skip map's size
            dY, dX = H2 - 2 * H, W2 - 2 * W
            pad = [dX // 2, dX - dX // 2, dY // 2, dY - dY // 2]
            u = torch.nn.functional.pad(u, pad).contiguous()
        sv, v = double_conv_fwd(p, cfg.pre + '.conv', [View(x2, x2.shape[1], H2, W2), View(u, C0, H2, W2)],
                                cfg.Co, n, gs, cfg.training, cfg.bufs, dev)
        out = _materialise(v, n, gs)
        ctx.cfg, ctx.p, ctx.sv, ctx.n, ctx.gs, ctx.x1, ctx.dims = cfg, p, sv, n, gs, x1, (Ci, C0, H, W, x2.shape[1])
        ctx.pad = pad
        return out

    @staticmethod
    def backward(ctx, dout):
        cfg, p, n, gs = ctx.cfg, ctx.p, ctx.n, ctx.gs
        Ci, C0, H, W, C2 = ctx.dims
        dev = dout.device
        lib = L.load()
        st = L.stream()
        grads = Grads(None, dev)
        dcat = double_conv_bwd(p, ctx.sv, n, gs, grads, dev, da=dout.contiguous(), dp=None)      # (n, C2 + C0, 2H, 2W)
        dx2 = dcat[:, :C2].contiguous()
        du = dcat[:, C2:]
        dus = dcat.stride(0)
        if ctx.pad is not None:                          # backward of F.pad: the un-padded window (negative pad = crop)
            du = torch.nn.functional.pad(du, [-v_ for v_ in ctx.pad]).contiguous()
            dus = du.stride(0)
        Wt = p[cfg.pre + '.up.weight']
        dWt = grads.buf(cfg.pre + '.up.weight', Wt.shape)
        ws = torch.empty(max(lib.mo_unet_wgrad_ws_floats(Ci, 4 * C0, n * H * W), n * C0 * 2), device=dev,
                         dtype=torch.float32)
        dbt = grads.buf(cfg.pre + '.up.bias', (C0,))
        L.call('mo_convt2x2_bwd_weight', du.data_ptr(), dus, C0, L.ptr(ctx.x1), Ci * H * W, Ci, None, None, 0, gs, n, H, W,
               L.ptr(dWt), L.ptr(dbt), L.ptr(ws), 0, st)
        dx1 = _empty(n, Ci, H, W, dev=dev)
        L.call('mo_convt2x2_bwd_data', du.data_ptr(), dus, C0, L.ptr(Wt), Ci, n, H, W, L.ptr(dx1), Ci * H * W, 0, st)
        return (None, dx1, dx2) + grads.result(cfg.names)


class OutConvFn(torch.autograd.Function):
    """unet.py:86-92: 1x1 conv with bias."""

    @staticmethod
    def forward(ctx, cfg, x, W, b):
        x = x.contiguous().float()
        n, Ci, H, Wd = x.shape
        Co = W.shape[0]
        out = _empty(n, Co, H, Wd, dev=x.device)
        L.call('mo_nchw_conv1x1_fwd', L.ptr(x), Ci * H * Wd, Ci, None, None, 0, 1, L.ptr(W), L.ptr(b), Co, n, H * Wd,
               L.ptr(out), Co * H * Wd, 0, L.stream())
        ctx.save_for_backward(x, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, W = ctx.saved_tensors
        dout = dout.contiguous()
        n, Ci, H, Wd = x.shape
        Co, HW = W.shape[0], H * Wd
        dev = x.device
        lib = L.load()
        dW = torch.empty_like(W)
        db = _empty(Co, dev=dev)
        ws = torch.empty(max(lib.mo_unet_wgrad_ws_floats(Co, Ci, n * HW), n * Co * 2), device=dev, dtype=torch.float32)
        L.call('mo_nchw_conv1x1_bwd_weight', L.ptr(dout), Co * HW, Co, L.ptr(x), Ci * HW, Ci, None, None, 0, 1, n, HW,
               L.ptr(dW), L.ptr(db), L.ptr(ws), 0, L.stream())
        dx = _empty(n, Ci, H, Wd, dev=dev)
        L.call('mo_nchw_conv1x1_bwd_data', L.ptr(dout), Co * HW, Co, L.ptr(W), Ci, n, HW, L.ptr(dx), Ci * HW, 0, L.stream())
        return None, dx, dW, db


class ContractionFn(torch.autograd.Function):
    """unet.py:106-126 over all counties at once: (n_img, Cin, S, S) -> x5 (n_img, 64, S/16, S/16) and the four
    activated skip maps x1..x4 (what the reference stashes in self.feature_maps)."""

    @staticmethod
    def forward(ctx, cfg, x, *params):
        p = dict(zip(cfg.names, params))
        x = x.contiguous().float()
        n, Cin, S, _ = x.shape
        dev = x.device
        gs = cfg.gsize
        saved = []
        sv, v = double_conv_fwd(p, 'inc', [View(x, Cin, S, S)], 4, n, gs, cfg.training, cfg.bufs, dev)
        saved.append(sv)
        views = [v]
        for k, (ci, co) in enumerate(ENC_CH, 1):
            pooled = _materialise(v, n, gs, pool=1)
            sv, v = double_conv_fwd(p, f'down{k}.maxpool_conv.1', [View(pooled, ci, v.H // 2, v.W // 2)], co, n, gs,
                                    cfg.training, cfg.bufs, dev)
            saved.append(sv)
            views.append(v)
        outs = tuple(_materialise(vv, n, gs) for vv in (views[4], views[0], views[1], views[2], views[3]))
        ctx.cfg, ctx.p, ctx.saved, ctx.n = cfg, p, saved, n
        ctx.need_dx = x.requires_grad
        return outs

    @staticmethod
    def backward(ctx, dx5, d1, d2, d3, d4):
        cfg, p, saved, n = ctx.cfg, ctx.p, ctx.saved, ctx.n
        dev = dx5.device
        grads = Grads(None, dev)
        dp = double_conv_bwd(p, saved[4], n, cfg.gsize, grads, dev, da=dx5.contiguous(), dp=None)
        dfm = [d1, d2, d3, d4]
        for k in (3, 2, 1, 0):
            da = dfm[k].contiguous() if dfm[k] is not None else None
            dp = double_conv_bwd(p, saved[k], n, cfg.gsize, grads, dev, da=da, dp=dp,
                                 need_input_grad=(k > 0) or ctx.need_dx)
        return (None, dp) + grads.result(cfg.names)


class FcBlockFn(torch.autograd.Function):
    """Encoder / Decoder (unet.py:138-149, 162-173): relu(fc1) -> dropout -> relu(fc2) on rows."""

    @staticmethod
    def forward(ctx, cfg, x, *params):
        p = dict(zip(cfg.names, params))
        x = x.contiguous().float()
        sv, out = fc_block_fwd(p, 'fc', x, _drop_params(cfg.drop_p, cfg.training))
        sv['h2'] = None                      # the output goes through save_for_backward (no ctx -> output cycle)
        ctx.save_for_backward(out)
        ctx.cfg, ctx.p, ctx.sv = cfg, p, sv
        ctx.need_dx = x.requires_grad
        return out

    @staticmethod
    def backward(ctx, dout):
        grads = Grads(None, dout.device)
        sv = dict(ctx.sv, h2=ctx.saved_tensors[0])
        dx = fc_block_bwd(ctx.p, sv, dout, grads, need_input_grad=ctx.need_dx)
        return (None, dx) + grads.result(ctx.cfg.names)


class ExpansionFn(torch.autograd.Function):
    """unet.py:184-199 over all counties: x (n_img, 64, s, s) and the four activated skip maps -> (n_img, Cout, S, S)."""

    @staticmethod
    def forward(ctx, cfg, x, f1, f2, f3, f4, *params):
        p = dict(zip(cfg.names, params))
        x = x.contiguous().float()
        n = x.shape[0]
        dev = x.device
        gs = cfg.gsize
        st = L.stream()
        skips = [View(f.contiguous().float(), f.shape[1], f.shape[2], f.shape[3]) for f in (f1, f2, f3, f4)]
        v = View(x, x.shape[1], x.shape[2], x.shape[3])
        ups = []
        for k, (ci, co) in enumerate(DEC_CH, 1):
            H = v.H
            u = _empty(n, ci // 2, 2 * H, 2 * H, dev=dev)
            L.call('mo_convt2x2_fwd', L.ptr(v.t), v.istride, ci, L.ptr(v.sc), L.ptr(v.sh), 1 if v.sc is not None else 0,
                   gs, L.ptr(p[f'up{k}.up.weight']), L.ptr(p[f'up{k}.up.bias']), ci // 2, n, H, H, L.ptr(u),
                   (ci // 2) * 4 * H * H, 0, st)
            sk = skips[4 - k]
            if sk.H != 2 * H:
                raise NotImplementedError('Up padding (unet.py:76-81) is only needed for odd sizes')
            sv, vn = double_conv_fwd(p, f'up{k}.conv', [sk, View(u, ci // 2, 2 * H, 2 * H)], co, n, gs, cfg.training,
                                     cfg.bufs, dev)
            ups.append(dict(vin=v, dc=sv, ci=ci, H=H))
            v = vn
        Wo, bo = p['outc.conv.weight'], p['outc.conv.bias']
        Cout = Wo.shape[0]
        out = _empty(n, Cout, v.H, v.W, dev=dev)
        L.call('mo_nchw_conv1x1_fwd', L.ptr(v.t), v.istride, v.C, L.ptr(v.sc), L.ptr(v.sh), 1, gs, L.ptr(Wo), L.ptr(bo),
               Cout, n, v.H * v.W, L.ptr(out), Cout * v.H * v.W, 0, st)
        ctx.cfg, ctx.p, ctx.ups, ctx.vlast, ctx.n = cfg, p, ups, v, n
        return out

    @staticmethod
    def backward(ctx, dout):
        cfg, p, n, v = ctx.cfg, ctx.p, ctx.n, ctx.vlast
        lib = L.load()
        gs = cfg.gsize
        dev = dout.device
        st = L.stream()
        grads = Grads(None, dev)
        dout = dout.contiguous()
        Wo = p['outc.conv.weight']
        Cout, C4 = Wo.shape[0], Wo.shape[1]
        HW = v.H * v.W
        dWo = grads.buf('outc.conv.weight', Wo.shape)
        dbo = grads.buf('outc.conv.bias', (Cout,))
        ws = torch.empty(max(lib.mo_unet_wgrad_ws_floats(Cout, C4, n * HW), n * Cout * 2), device=dev, dtype=torch.float32)
        L.call('mo_nchw_conv1x1_bwd_weight', L.ptr(dout), Cout * HW, Cout, L.ptr(v.t), v.istride, C4, L.ptr(v.sc),
               L.ptr(v.sh), 1, gs, n, HW, L.ptr(dWo), L.ptr(dbo), L.ptr(ws), 0, st)
        da = _empty(n, C4, v.H, v.W, dev=dev)
        L.call('mo_nchw_conv1x1_bwd_data', L.ptr(dout), Cout * HW, Cout, L.ptr(Wo), C4, n, HW, L.ptr(da), C4 * HW, 0, st)
        dfm = [None] * 4
        for k in (4, 3, 2, 1):
            up = ctx.ups[k - 1]
            ci, H, vin = up['ci'], up['H'], up['vin']
            dcat = double_conv_bwd(p, up['dc'], n, gs, grads, dev, da=da, dp=None)
            C0 = ci // 2
            dfm[4 - k] = dcat[:, :C0].contiguous()
            du = dcat[:, C0:]
            dus = dcat.stride(0)
            Wt = p[f'up{k}.up.weight']
            dWt = grads.buf(f'up{k}.up.weight', Wt.shape)
            wsu = torch.empty(max(lib.mo_unet_wgrad_ws_floats(ci, 4 * C0, n * H * H), n * C0 * 2), device=dev,
                              dtype=torch.float32)
            dbt = grads.buf(f'up{k}.up.bias', (C0,))
            L.call('mo_convt2x2_bwd_weight', du.data_ptr(), dus, C0, L.ptr(vin.t), vin.istride, ci, L.ptr(vin.sc),
                   L.ptr(vin.sh), 1 if vin.sc is not None else 0, gs, n, H, H, L.ptr(dWt), L.ptr(dbt), L.ptr(wsu), 0, st)
            da = _empty(n, ci, H, H, dev=dev)
            L.call('mo_convt2x2_bwd_data', du.data_ptr(), dus, C0, L.ptr(Wt), ci, n, H, H, L.ptr(da), ci * H * H, 0, st)
        return (None, da, dfm[0], dfm[1], dfm[2], dfm[3]) + grads.result(cfg.names)


# ---------------------------------------------------------------------------------------------- module-level helpers
def _named(mod):
    names = [k for k, _ in mod.named_parameters()]
    return names, [p for _, p in mod.named_parameters()]


def _bn_bufs(mod):
    return {name: (m.running_mean, m.running_var, m.num_batches_tracked)
            for name, m in mod.named_modules() if isinstance(m, nn.BatchNorm2d)}


def double_conv_forward(mod, x, pool=False):
    """DoubleConv.forward / Down.forward: `mod` is the DoubleConv (keys double_conv.{0,1,3,4}.*)."""
    _need_gpu(x)
    names, params = _named(mod)
    bufs = {'m.' + k: v for k, v in _bn_bufs(mod).items()}
    cfg = _Cfg(['m.' + k for k in names], bufs, mod.training, pre='m', pool=pool, gsize=None,
               Co=mod.double_conv[0].out_channels)
    return DoubleConvFn.apply(cfg, x, *params)


def up_forward(mod, x1, x2):
    _need_gpu(x1)
    names, params = _named(mod)
    bufs = {'m.' + k: v for k, v in _bn_bufs(mod).items()}
    cfg = _Cfg(['m.' + k for k in names], bufs, mod.training, pre='m', gsize=None,
               Co=mod.conv.double_conv[0].out_channels)
    return UpFn.apply(cfg, x1, x2, *params)


def outconv_forward(mod, x):
    _need_gpu(x)
    return OutConvFn.apply(None, x, mod.conv.weight, mod.conv.bias)


def contraction_forward(mod, inp):
    """Contraction.forward (unet.py:106-126): inp (n_counties, H, Cin, S, S) -> (n_counties, H, 64*(S/16)^2); the four
    skip maps are left in mod.feature_maps as (n_counties, H, C, h, w) tensors, as the reference's torch.stack does."""
    _need_gpu(inp)
    NC, H, Cin, S, _ = inp.shape
    names, params = _named(mod)
    cfg = _Cfg(names, _bn_bufs(mod), mod.training, gsize=H)
    outs = ContractionFn.apply(cfg, inp.reshape(NC * H, Cin, S, S), *params)
    mod.feature_maps = [o.view(NC, H, *o.shape[1:]) for o in outs[1:]]
    return outs[0].reshape(NC, mod.horizon, -1)


def fc_forward(mod, inp):
    """Encoder.forward / Decoder.forward on (n_counties, H, F) rows (dropout1.p, train mode only)."""
    _need_gpu(inp)
    NC, H, Fin = inp.shape
    cfg = _Cfg(['fc.fc1.weight', 'fc.fc1.bias', 'fc.fc2.weight', 'fc.fc2.bias'], None, mod.training,
               drop_p=mod.dropout1.p)
    out = FcBlockFn.apply(cfg, inp.reshape(NC * H, Fin), mod.fc1.weight, mod.fc1.bias, mod.fc2.weight, mod.fc2.bias)
    return out.view(NC, H, -1)


def expansion_forward(mod, inp, feature_maps):
    """Expansion.forward (unet.py:184-199): inp (n_counties, H, 64, s, s), feature_maps = the four stacked skip maps of
    Contraction (indexable per county, each (H, C, h, w))."""
    _need_gpu(inp)
    NC, H = inp.shape[0], inp.shape[1]
    names, params = _named(mod)
    fms = [fm if torch.is_tensor(fm) else torch.stack(list(fm)) for fm in feature_maps]
    fms = [fm.reshape(NC * H, *fm.shape[2:]) for fm in fms]
    cfg = _Cfg(names, _bn_bufs(mod), mod.training, gsize=H)
    out = ExpansionFn.apply(cfg, inp.reshape(NC * H, *inp.shape[2:]), *fms, *params)
    return out.view(NC, H, *out.shape[1:])
