"""Host-side orchestration of the UNet encoder/decoder (unet.py:95-199) on the HIP C-ABI.

All B*n_counties*H tiles are processed in one batch of NCHW images (image index = (b*n_counties +
county)*H + h); BatchNorm runs per group of H images, i.e. exactly the statistics of the reference's
per-county calls, and running stats receive the same sequential county-then-batch updates
(SURVEY.md F7).  Two autograd Functions: ``UnetEncodeFn`` (Contraction + Encoder) and
``UnetDecodeFn`` (Decoder + Expansion); the skip feature maps travel between them as raw conv
outputs whose "gradient" is defined as the gradient w.r.t. their activated view.
"""
import os

import torch

from . import _lib as L

WGRAD_LANE = os.environ.get('MO_UNET_WGRAD_LANE', '1') != '0'     # A/B switch: weight gradients on a side stream
# experiments on where the bf16 mode's deep-stage gradient error comes from (tests: modified_unet_H7 by stage)
GRAD_F32 = os.environ.get('MO_UNET_GRAD_F32', '0') != '0'         # gradient tensors (dy, da, dx) stored fp32 in the bf16 mode
DGRAD_F32 = os.environ.get('MO_UNET_DGRAD_F32', '0') != '0'       # data gradients on the exact-fp32 kernels in the bf16 mode
U_BF = os.environ.get('MO_UNET_U_BF', '1') != '0'                 # A/B switch: upsampled maps (and the concat gradient) of up3 / up4 as bf16


class _Lane:
    """Weight / bias-gradient kernels are off the data-flow chain of the backward pass (activation backward -> data
    gradient -> next layer): they run on a side HIP stream beside it and are joined at the end of the Function's
    backward.  The kernels get the side stream's handle directly (no torch stream-context switch: ~20 us of host time
    per call, 45 calls per step), and every tensor they touch -- operands allocated on the main stream, workspaces -- is
    simply kept alive until join(): after the main stream has waited for the lane, the allocator may recycle them."""
    _streams = {}
    _deferred = {}

    def __init__(self, dev, enabled=True):
        self.main = torch.cuda.current_stream()
        self.side = None
        self.held = []
        if enabled and WGRAD_LANE:
            key = self.key = (torch.device(dev).index, 'unet_wgrad')
            if key not in _Lane._streams:
                _Lane._streams[key] = torch.cuda.Stream(device=dev)
            self.side = _Lane._streams[key]
            self.side_handle = self.side.cuda_stream

    def keep(self, t):
        """A tensor the lane's kernels use (workspace): alive until join()."""
        if self.side is not None:
            self.held.append(t)
        return t

    def run(self, fn, reads=()):
        """fn(stream_handle) launches on the lane behind everything queued on the main stream so far."""
        if self.side is None:
            fn(L.stream())
            return
        self.side.wait_stream(self.main)
        self.held.extend(reads)
        fn(self.side_handle)

    def join(self):
        if self.side is not None:
            self.main.wait_stream(self.side)
            self.held.clear()
            _Lane._deferred.pop(self.key, None)

    def defer(self):
        """Instead of join(): leave the lane's work running beside whatever the main stream does next and hand the tensors
        it uses to the NEXT join() on this side stream.  Only for a backward whose lane results (in-place parameter
        gradients) have no consumer on the main stream before that join -- the decoder's backward inside Modified_UNET,
        which the Graph WaveNet's and the contraction's backward follow (the main stream idled ~0.3 ms of a 7 ms step
        waiting for the decoder's last weight gradients and their Adam update)."""
        if self.side is not None:
            _Lane._deferred.setdefault(self.key, []).extend(self.held)
            self.held = []


def join_deferred():
    """The current stream waits for every weight-gradient lane whose join was deferred (_Lane.defer) and has not happened
    since -- called by the trainer in front of the collectives / the optimizer step, for graphs in which the backward
    that would have joined the lane did not run (a frozen contraction)."""
    for key in list(_Lane._deferred):
        torch.cuda.current_stream().wait_stream(_Lane._streams[key])
        _Lane._deferred.pop(key, None)


ENC_CH = ((4, 8), (8, 16), (16, 32), (32, 64))      # down1..4 (unet.py:100-103)
DEC_CH = ((64, 32), (32, 16), (16, 8), (8, 4))      # up1..4   (unet.py:178-181)


class View:
    """An activated view: raw NCHW tensor + per-(group,channel) folded BatchNorm affine + ReLU."""

    def __init__(self, t, C, H, W, sc=None, sh=None, off=None):
        self.t, self.C, self.H, self.W, self.sc, self.sh = t, C, H, W, sc, sh
        self.off = off        # optional int64 per-image element offsets (the network input as a permuted view, lit.py:31)

    @property
    def istride(self):
        return self.C * self.H * self.W

    @property
    def bf(self):
        """1 when the tensor is stored as bf16 ("bf16 mode": raw conv outputs and gradients at the large resolutions)."""
        return int(self.t.dtype == torch.bfloat16)

    def args(self):
        return (self.t.data_ptr() if self.off is not None else L.ptr(self.t), self.C, self.istride, L.ptr(self.sc),
                L.ptr(self.sh), 1 if self.sc is not None else 0)


_NOVIEW = (None, 0, 0, None, None, 0)


def _empty(*shape, dev, bf=False):
    return torch.empty(shape, device=dev, dtype=torch.bfloat16 if bf else torch.float32)


def bf_ok(mode, Co, H, W):
    """bf16 storage of an activation tensor of Co channels at H x W: only where every kernel that touches it is a direct /
    streaming one (include/mo_hip.h `dtypes`): the MFMA weight gradient needs W % 64 == 0 and H % 8 == 0, the direct
    conv <= 32 output channels."""
    return bool(mode == 'bf16' and Co <= 32 and H % 8 == 0 and W % 64 == 0 and H >= 64)      # (narrower levels stay fp32)


def _is_bf(t):
    return int(t is not None and t.dtype == torch.bfloat16)


class Grads(dict):
    """Gradient destinations of one backward pass: the trainer's flat-buffer view when one is registered for the key
    (state['grad_out'], written in place and NOT handed to autograd), else a new tensor."""

    def __init__(self, gout, dev):
        super().__init__()
        self.gout, self.dev = gout or {}, dev

    def buf(self, key, shape):
        t = self.gout.get(key)
        if t is None:
            t = torch.empty(tuple(shape), device=self.dev, dtype=torch.float32)
        self[key] = t
        return t

    def result(self, names):
        return tuple(None if (k in self.gout and self.gout[k] is not None) else self.get(k) for k in names)


def _conv_bn(p, wkey, bnkey, views, Co, n, gs, training, bufs, dev, out_bf=False, math=False):
    v0 = views[0]
    H, W = v0.H, v0.W
    st = L.stream()
    y = _empty(n, Co, H, W, dev=dev, bf=out_bf)
    a1 = views[1].args() if len(views) > 1 else _NOVIEW
    dt = (L.BF_IN0 * v0.bf) | (L.BF_IN1 * (views[1].bf if len(views) > 1 else 0)) | (L.BF_OUT * int(out_bf)) | \
         (L.BF_MATH * int(math))
    # train mode: the BatchNorm statistics come out of the conv's own epilogue where the direct kernels run
    # (per-tile partial sums), else from one pass over the output
    ntile = (L.load().mo_conv3x3_stats_tiles2(v0.C, views[1].C if len(views) > 1 else 0, Co, n, H, W, dt)
             if training else 0)
    stats = _empty(n, ntile, Co, 2, dev=dev) if ntile else None
    L.call('mo_conv3x3_fwd', *v0.args(), *a1, gs, L.ptr(p[wkey]), Co, n, H, W, L.ptr(y), Co * H * W, L.ptr(stats), dt, L.ptr(v0.off), st)
    G = n // gs
    aff = _empty(4, G, Co, dev=dev)            # scale, shift, mean, rstd
    if training and not ntile:
        stats = _empty(n, Co, 2, dev=dev)
        L.call('mo_nchw_stats', L.ptr(y), Co * H * W, Co, n, H * W, L.ptr(stats), st)
    rm, rv, nbt = bufs[bnkey]
    # (train mode: statistics, folded affine, the G sequential running-stat updates and num_batches_tracked in one launch)
    L.call('mo_group_bn_finalize2', L.ptr(stats), n, Co, gs, H * W, max(ntile, 1), L.ptr(p[bnkey + '.weight']),
           L.ptr(p[bnkey + '.bias']), L.ptr(rm), L.ptr(rv), 0.1, 1e-5, 1 if training else 0,
           L.ptr(aff[0]), L.ptr(aff[1]), L.ptr(aff[2]), L.ptr(aff[3]), nbt.data_ptr() if training else None, st)
    return y, aff


def double_conv_fwd(p, pre, views, Co, n, gs, training, bufs, dev, bf=(False, False), math=False):
    """unet.py:40-53.  Returns (saved, output view).  bf = (y1, y2 stored as bf16); math: the convs (and their gradients)
    on the bf16 matrix pipe where a kernel exists (MO_BF_MATH; fp32 accumulation)."""
    H, W = views[0].H, views[0].W
    y1, aff1 = _conv_bn(p, pre + '.double_conv.0.weight', pre + '.double_conv.1', views, Co, n, gs, training, bufs, dev,
                        bf[0], math)
    v1 = View(y1, Co, H, W, aff1[0], aff1[1])
    y2, aff2 = _conv_bn(p, pre + '.double_conv.3.weight', pre + '.double_conv.4', [v1], Co, n, gs, training, bufs, dev,
                        bf[1], math)
    v2 = View(y2, Co, H, W, aff2[0], aff2[1])
    return dict(pre=pre, views=views, y1=y1, aff1=aff1, v1=v1, y2=y2, aff2=aff2, Co=Co, H=H, W=W, math=math), v2


def _flip(W, dev):
    Co, Ci = W.shape[0], W.shape[1]
    Wf = _empty(Ci, Co, 3, 3, dev=dev)
    L.call('mo_conv3x3_flip_weights', L.ptr(W), Co, Ci, L.ptr(Wf), L.stream())
    return Wf


def _dastride(t):
    """Image stride of a gradient tensor that is contiguous within each image (channel slices allowed)."""
    if t is None:
        return None, 0
    n, C, H, W = t.shape
    assert t.stride(3) == 1 and t.stride(2) == W and t.stride(1) == H * W, 'gradient must be image-contiguous'
    return t, t.stride(0)


def double_conv_bwd(p, sv, n, gs, grads, dev, da=None, dp=None, need_input_grad=True, dx_bf=False, lane=None,
                    da_scale=None):
    """Backward of DoubleConv.  da: gradient w.r.t. the activated output view (may be a channel slice of
    a wider buffer), dp: gradient w.r.t. its 2x2 max-pooled version.  Returns the gradient w.r.t. the
    (activated) channel-concatenated input, shape (n, C0+C1, H, W), or None."""
    lib = L.load()
    st = L.stream()
    pre, Co, H, W = sv['pre'], sv['Co'], sv['H'], sv['W']
    HW = H * W
    math = L.BF_MATH * int(sv.get('math', False))

    def act_bwd(y, aff, bnkey, da_t, dp_t, scale=None):
        dy = _empty(n, Co, H, W, dev=dev, bf=_is_bf(y) and not GRAD_F32)    # the gradient of a conv output is stored as the output is
        dg = grads.buf(bnkey + '.weight', (Co,))
        db = grads.buf(bnkey + '.bias', (Co,))
        ws = torch.empty(lib.mo_unet_act_bwd_ws_floats(n, Co), device=dev, dtype=torch.float32)
        da_t, das = _dastride(da_t)
        L.call('mo_unet_act_bwd', L.ptr(y), Co * HW, Co, n, H, W, gs, L.ptr(p[bnkey + '.weight']), L.ptr(aff[2]),
               L.ptr(aff[3]), L.ptr(aff[0]), L.ptr(aff[1]), da_t.data_ptr() if da_t is not None else None, das,
               L.ptr(dp_t), (Co * HW) // 4, L.ptr(dy), Co * HW, L.ptr(dg), L.ptr(db), L.ptr(ws),
               (L.BF_IN0 * _is_bf(y)) | (L.BF_IN1 * _is_bf(da_t)) | (L.BF_DP * _is_bf(dp_t)) | (L.BF_OUT * _is_bf(dy)),
               L.ptr(scale), st)
        return dy

    lane = lane or _Lane(dev, False)

    def wgrad(dy, views, wkey):
        Ci = sum(v.C for v in views)
        dW = grads.buf(wkey, (Co, Ci, 3, 3))
        a1 = views[1].args() if len(views) > 1 else _NOVIEW
        dt = (L.BF_DY * _is_bf(dy)) | (L.BF_IN0 * views[0].bf) | (L.BF_IN1 * (views[1].bf if len(views) > 1 else 0)) | math

        def fn(ls):
            ws = lane.keep(torch.empty(lib.mo_unet_wgrad_ws_floats(Co, Ci * 9, n * HW), device=dev, dtype=torch.float32))
            L.call('mo_conv3x3_bwd_weight', L.ptr(dy), Co * HW, Co, *views[0].args(), *a1, gs, n, H, W, L.ptr(dW),
                   L.ptr(ws), dt, L.ptr(views[0].off), ls)
        lane.run(fn, reads=[dy, dW] + [t for v in views for t in (v.t, v.sc, v.sh)])

    def dgrad(dy, Wt, out_bf):
        Ci = Wt.shape[1]
        out_bf = bool(out_bf) and not GRAD_F32
        dx = _empty(n, Ci, H, W, dev=dev, bf=out_bf)
        dt = (L.BF_IN0 * _is_bf(dy)) | (L.BF_OUT * int(out_bf))
        if math and not DGRAD_F32 and lib.mo_conv3x3_bf16_route(Co, Ci, n, H, W):
            # the bf16 matrix-pipe kernel reads the forward weights transposed + flipped in place
            L.call('mo_conv3x3_fwd', L.ptr(dy), Co, Co * HW, None, None, 0, *_NOVIEW, 1, L.ptr(Wt), Ci, n, H, W,
                   L.ptr(dx), Ci * HW, None, dt | math | L.W_FLIP, None, st)
            return dx
        Wf = _flip(Wt, dev)
        L.call('mo_conv3x3_fwd', L.ptr(dy), Co, Co * HW, None, None, 0, *_NOVIEW, 1, L.ptr(Wf), Ci, n, H, W,
               L.ptr(dx), Ci * HW, None, dt, None, st)
        return dx

    # da_scale (device scalar): `da` was formed for an upstream loss gradient of 1 (mo_outc_loss_fwd); everything behind
    # this activation backward is linear in it, so it is applied once, here
    dy2 = act_bwd(sv['y2'], sv['aff2'], pre + '.double_conv.4', da, dp, da_scale)
    wgrad(dy2, [sv['v1']], pre + '.double_conv.3.weight')
    da1 = dgrad(dy2, p[pre + '.double_conv.3.weight'], _is_bf(sv['y1']))      # consumed by y1's activation backward
    dy1 = act_bwd(sv['y1'], sv['aff1'], pre + '.double_conv.1', da1, None)
    wgrad(dy1, sv['views'], pre + '.double_conv.0.weight')
    if not need_input_grad:
        return None
    return dgrad(dy1, p[pre + '.double_conv.0.weight'], dx_bf)


# bf16 mode: Linear layers against weight matrices of at least this many elements (Encoder.fc1 16384 -> 4096 and Decoder.fc2
# 1024 -> 16384 of config 3) run as "3 x bf16" split products on the bf16 matrix pipe (mo_fc3_*: ~1.5e-5 relative per product,
# 3x faster than the exact-fp32 MFMA, which is compute bound on them).  MO_FC3=0: exact fp32 everywhere (A/B switch).
# (A plain bf16 product was measured too: 0.3 ms faster still, but it moved the deep-stage gradient distance of the config-3
# golden from 0.69 to 0.78 -- the 3-way split leaves it where it was.)
FC3_MIN = (1 << 22) if os.environ.get('MO_FC3', '1') != '0' else (1 << 62)


def _fc_fwd(x, W, b, relu, math=False):
    P, Ci = x.shape
    Co = W.shape[0]
    out = _empty(P, Co, dev=x.device)
    if math and W.numel() >= FC3_MIN and L.load().mo_fc3_supported(P, Ci, Co):
        # (more rows than one window's 134: the kernel walks row groups of 144 itself, its weight panel re-read from L2)
        ws = torch.empty(L.load().mo_fc3_ws_floats(P, Ci, Co), device=x.device, dtype=torch.float32)
        L.call('mo_fc3_fwd', L.ptr(x), P, Ci, L.ptr(W), L.ptr(b), Co, 1 if relu else 0, L.ptr(out), L.ptr(ws), L.stream())
        return out
    if Ci >= 2048 and P <= 1024:        # few rows x long K: split-K (the tile grid alone would be ~20 workgroups)
        ws = torch.empty(L.load().mo_linear_splitk_ws_floats(P, Co, Ci), device=x.device, dtype=torch.float32)
        L.call('mo_conv1x1_fwd_splitk', L.ptr(x), Ci, L.ptr(W), L.ptr(b), Co, L.ptr(out), P, 1 if relu else 0,
               L.ptr(ws), L.stream())
        return out
    L.call('mo_conv1x1_fwd', L.ptr(x), Ci, 0, 0, 0, 0, L.ptr(W), L.ptr(b), Co, L.ptr(out), P, 1 if relu else 0, 0,
           L.stream())
    return out


def _dropout(x, seed, thresh, scale):
    if not thresh:
        return x
    y = torch.empty_like(x)
    L.call('mo_dropout', L.ptr(x), L.ptr(y), x.numel(), seed, thresh, scale, L.stream())
    return y


def fc_block_fwd(p, pre, x, drop, math=False):
    """unet.py:138-149 / :162-173: relu(fc1) -> dropout -> relu(fc2) on rows of x.  math (bf16 mode): the products against
    the large weight matrices as 3 x bf16 split products (FC3_MIN above)."""
    h1 = _fc_fwd(x, p[pre + '.fc1.weight'], p[pre + '.fc1.bias'], True, math)
    d1 = _dropout(h1, *drop)
    h2 = _fc_fwd(d1, p[pre + '.fc2.weight'], p[pre + '.fc2.bias'], True, math)
    return dict(pre=pre, x=x, h1=h1, d1=d1, h2=h2, drop=drop, math=math), h2


def fc_block_bwd(p, sv, dh2, grads, need_input_grad=True, lane=None):
    lib = L.load()
    st = L.stream()
    pre, x, h1, d1, h2 = sv['pre'], sv['x'], sv['h1'], sv['d1'], sv['h2']
    dev = x.device
    P = x.shape[0]
    lane = lane or _Lane(dev, False)

    def lin_bwd(dout, inp, wkey, bkey, need_in):
        W = p[wkey]
        Co, Ci = W.shape
        dW = grads.buf(wkey, (Co, Ci))
        db = grads.buf(bkey, (Co,))

        def fn(ls):
            if sv.get('math') and W.numel() >= FC3_MIN and P <= 160:
                ws = lane.keep(torch.empty(lib.mo_fc3_wgrad_ws_floats(P, Co, Ci), device=dev, dtype=torch.float32))
                L.call('mo_fc3_bwd_weight', L.ptr(dout), P, Co, L.ptr(inp), Ci, L.ptr(dW), L.ptr(db), L.ptr(ws), ls)
                return
            ws = lane.keep(torch.empty(lib.mo_wgrad_ws_floats(Co, Ci, P), device=dev, dtype=torch.float32))
            L.call('mo_conv1x1_bwd_weight', L.ptr(dout), Co, P, L.ptr(inp), Ci, 0, 0, 0, 0, L.ptr(dW), L.ptr(db), L.ptr(ws),
                   ls)
        lane.run(fn, reads=[dout, inp, dW, db])
        if not need_in:
            return None
        din = _empty(P, Ci, dev=dev)
        if sv.get('math') and W.numel() >= FC3_MIN and lib.mo_fc3_supported(P, Co, Ci):
            wsd = torch.empty(lib.mo_fc3_ws_floats(P, Co, Ci), device=dev, dtype=torch.float32)
            L.call('mo_fc3_bwd_data', L.ptr(dout), P, Co, L.ptr(W), Ci, L.ptr(din), L.ptr(wsd), st)
            return din
        if Co >= 2048 and P <= 1024:
            wsd = torch.empty(lib.mo_linear_splitk_ws_floats(P, Ci, Co), device=dev, dtype=torch.float32)
            L.call('mo_conv1x1_bwd_data_splitk', L.ptr(dout), Co, P, L.ptr(W), Ci, L.ptr(din), L.ptr(wsd), st)
            return din
        L.call('mo_conv1x1_bwd_data', L.ptr(dout), Co, P, L.ptr(W), Ci, L.ptr(din), 0, 0, 0, None, 0, st)
        return din

    dh2 = dh2.contiguous()
    da2 = torch.empty_like(dh2)
    L.call('mo_relu_bwd', L.ptr(dh2), L.ptr(h2), L.ptr(da2), da2.numel(), st)
    dd1 = lin_bwd(da2, d1, pre + '.fc2.weight', pre + '.fc2.bias', True)
    dh1 = _dropout(dd1, *sv['drop'])
    da1 = torch.empty_like(dh1)
    L.call('mo_relu_bwd', L.ptr(dh1), L.ptr(h1), L.ptr(da1), da1.numel(), st)
    return lin_bwd(da1, x, pre + '.fc1.weight', pre + '.fc1.bias', need_input_grad)


def _drop_params(p_drop, training):
    if not training or p_drop <= 0:
        return (0, 0, 1.0)
    seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    return (seed, int(min(p_drop, 0.999999) * 4294967296.0), 1.0 / (1.0 - p_drop))


class UnetEncodeFn(torch.autograd.Function):
    """Contraction + Encoder (unet.py:106-126, 138-149) over all tiles.
    x: (n_img, Cin, S, S) -> feat (n_img, 256), fm1..fm4 (raw second-conv outputs of inc, down1..3)."""

    @staticmethod
    def forward(ctx, state, x, *params):
        p = dict(zip(state['names'], params))
        dev = x.device
        gs, training, bufs = state['gsize'], state['training'], state['bufs']
        x_off = None
        if x.dim() == 6:
            # the (B, NC, H, Cin, S, S) batch as a permuted view (lit.py:31): read in place through per-image offsets
            n, (Cin, S) = x.shape[0] * x.shape[1] * x.shape[2], x.shape[3:5]
            x_off = state['x_off']
            ctx.x_shape = tuple(x.shape)
        else:
            n, Cin, S, _ = x.shape
            x = x.contiguous()
            ctx.x_shape = None
        st = L.stream()
        saved = []
        mode = state.get('act_dtype', 'f32')      # 'bf16': conv outputs / gradients of the large levels stored as bf16
        b0 = bf_ok(mode, 4, S, S)
        math = mode == 'bf16'
        sv, v = double_conv_fwd(p, 'contraction.inc', [View(x, Cin, S, S, off=x_off)], 4, n, gs, training, bufs, dev,
                                bf=(b0, b0), math=math)
        saved.append(sv)
        views = [v]
        for k, (ci, co) in enumerate(ENC_CH, 1):
            H = v.H
            bk = bf_ok(mode, co, H // 2, H // 2)
            pooled = _empty(n, ci, H // 2, H // 2, dev=dev, bf=bk)
            L.call('mo_unet_act', L.ptr(v.t), v.istride, ci, n, H, H, L.ptr(v.sc), L.ptr(v.sh), gs, 1, L.ptr(pooled),
                   ci * (H // 2) * (H // 2), (L.BF_IN0 * v.bf) | (L.BF_OUT * int(bk)), st)
            sv, v = double_conv_fwd(p, f'contraction.down{k}.maxpool_conv.1', [View(pooled, ci, H // 2, H // 2)], co, n,
                                    gs, training, bufs, dev, bf=(bk, bk), math=math)
            saved.append(sv)
            views.append(v)
        v5 = views[-1]
        x5a = _empty(n, v5.C, v5.H, v5.W, dev=dev)
        L.call('mo_unet_act', L.ptr(v5.t), v5.istride, v5.C, n, v5.H, v5.W, L.ptr(v5.sc), L.ptr(v5.sh), gs, 0, L.ptr(x5a),
               v5.istride, L.BF_IN0 * v5.bf, st)
        fc_sv, feat = fc_block_fwd(p, 'encoder', x5a.view(n, -1), _drop_params(state['fc_dropout'], training), math=math)
        # the decoder needs the skip maps' folded BatchNorm affine; the maps themselves reach it as Function inputs
        state['skip_meta'] = [(v.C, v.H, v.W, v.sc, v.sh) for v in views[:4]]
        # tensors that are OUTPUTS of this Function must not be referenced from ctx attributes: output -> grad_fn ->
        # ctx -> output is a reference cycle, the step's activations then live until Python's cyclic collector runs
        # and the caching allocator has to grow (hipMalloc) in the middle of later steps -- the multi-second stall of
        # round 1's UNet bench leg.  They go through save_for_backward and are put back in backward.
        outs = (feat,) + tuple(v.t for v in views[:4])
        ctx.save_for_backward(*outs)
        if state.get('skip_side') is not None:
            # Modified_UNET: the skip maps' gradients come back through state['skip_side'] (set by UnetDecodeFn.backward,
            # which always runs first: feat -> st_gnn -> z -> decoder), not through autograd -- which would cast the fp32
            # channel slices of the decoder's concat gradient to the maps' bf16 (four strided copies, 250 MB per window)
            ctx.mark_non_differentiable(*outs[1:])
            ctx.set_materialize_grads(False)
        fc_sv['h2'] = None
        for k in range(4):
            saved[k]['y2'] = None
        ctx.state, ctx.p, ctx.saved, ctx.fc_sv, ctx.n = state, p, saved, fc_sv, n
        ctx.x_needs_grad = x.requires_grad
        return outs

    @staticmethod
    def backward(ctx, dfeat, dfm1, dfm2, dfm3, dfm4):
        state, p, saved, n = ctx.state, ctx.p, ctx.saved, ctx.n
        gs = state['gsize']
        dev = dfeat.device
        grads = Grads(state.get('grad_out'), dev)
        outs = ctx.saved_tensors
        fc_sv = dict(ctx.fc_sv, h2=outs[0])
        saved = [dict(sv, y2=outs[1 + k]) if k < 4 else sv for k, sv in enumerate(saved)]
        lane = _Lane(dev)
        dx5a = fc_block_bwd(p, fc_sv, dfeat, grads, lane=lane)
        if state.get('adam_now') is not None and state.get('grad_out'):
            # a flat-buffer trainer in eager mode: the FC bottleneck's and the Graph WaveNet's gradients are final (the
            # decoder side's were announced at the end of its backward) -- their Adam update goes on the weight-gradient
            # lane now and runs beside the contraction's backward
            pre = ('encoder.', 'st_gnn.') if state.get('st_gnn_in_place') else ('encoder.',)
            lane.run(lambda ls: state['adam_now'](pre, ls))
        v5 = saved[4]
        dp = double_conv_bwd(p, saved[4], n, gs, grads, dev, da=dx5a.view(n, v5['Co'], v5['H'], v5['W']), dp=None,
                             dx_bf=bool(saved[4]['views'][0].bf), lane=lane)
        dfm = [dfm1, dfm2, dfm3, dfm4]
        side = state.get('skip_side')
        if side is not None:
            dfm = side.pop('dfm', dfm)
        for k in (3, 2, 1, 0):
            need = (k > 0) or ctx.x_needs_grad
            # the gradient w.r.t. a level's pooled input is the dp of the level above: stored as that input is
            dp = double_conv_bwd(p, saved[k], n, gs, grads, dev, da=dfm[k], dp=dp, need_input_grad=need,
                                 dx_bf=bool(k > 0 and saved[k]['views'][0].bf), lane=lane)
        lane.join()
        if dp is not None and ctx.x_shape is not None:
            dp = dp.view(ctx.x_shape)
        # the step's activations hang on this node as attributes: release them now, not when the caller drops the loss
        ctx.saved = ctx.fc_sv = None
        return (None, dp) + grads.result(state['names'])


class UnetDecodeFn(torch.autograd.Function):
    """Decoder + Expansion (unet.py:162-173, 184-199).  z: (n_img, 256) and the four skip maps."""

    @staticmethod
    def forward(ctx, state, z, fm1, fm2, fm3, fm4, *params):
        p = dict(zip(state['names'], params))
        dev = z.device
        n = z.shape[0]
        gs, training, bufs = state['gsize'], state['training'], state['bufs']
        st = L.stream()
        fc_sv, h2 = fc_block_fwd(p, 'decoder', z.contiguous(), _drop_params(state['fc_dropout'], training),
                                 math=state.get('act_dtype', 'f32') == 'bf16')
        skips = [View(fm, *meta) for fm, meta in zip((fm1, fm2, fm3, fm4), state['skip_meta'])]
        S0 = skips[3].H // 2
        v = View(h2.view(n, 64, S0, S0), 64, S0, S0)
        ups = []
        for k, (ci, co) in enumerate(DEC_CH, 1):
            H = v.H
            # bf16 mode, the levels whose ConvTranspose2d runs on the streaming kernels (up3, up4): the upsampled map -- and in backward the concat gradient behind it -- is
            # stored as bf16.  The concat conv and its weight gradient round their operands to bf16 on the way into the
            # MFMA anyway, so the forward result is bit-identical; per window this takes 0.37 GB (u: one write, two reads)
            # and 0.6 GB (the gradient: written by the concat conv's data gradient, read by the skip side's activation
            # backward and by ConvTranspose2d's two gradients) off the step's 18 GB.
            ubf = bool(U_BF and bf_ok(state.get('act_dtype', 'f32'), co, 2 * H, 2 * H) and not GRAD_F32 and
                       L.load().mo_convt2x2_bf16_route(ci, ci // 2, n))
            u = _empty(n, ci // 2, 2 * H, 2 * H, dev=dev, bf=ubf)
            L.call('mo_convt2x2_fwd', L.ptr(v.t), v.istride, ci, L.ptr(v.sc), L.ptr(v.sh), 1 if v.sc is not None else 0,
                   gs, L.ptr(p[f'expansion.up{k}.up.weight']), L.ptr(p[f'expansion.up{k}.up.bias']), ci // 2, n, H, H,
                   L.ptr(u), (ci // 2) * 4 * H * H, (L.BF_IN0 * v.bf) | (L.BF_OUT * int(ubf)), st)
            sk = skips[4 - k]
            if sk.H != 2 * H:
                raise NotImplementedError('Up padding (unet.py:76-81) inside Modified_UNET: only reached by image sizes that '
                                          'are not multiples of 16, whose odd widths the conv kernels (W % 4 == 0) do '
                                          'not serve; the stand-alone Up block pads (unet_blocks.UpFn)')
            # bf16 mode: y1 of an Up block and the last block's y2 (read by OutConv's streaming kernels); the y2 of up1..3
            # feeds the next ConvTranspose2d, which runs on the fp32 tile engine, and stays fp32 (as does u)
            bk = bf_ok(state.get('act_dtype', 'f32'), co, 2 * H, 2 * H)
            sv, vn = double_conv_fwd(p, f'expansion.up{k}.conv', [sk, View(u, ci // 2, 2 * H, 2 * H)], co, n, gs,
                                     training, bufs, dev, bf=(bk, bk and k == 4),
                                     math=state.get('act_dtype', 'f32') == 'bf16')
            ups.append(dict(vin=v, dc=sv, ci=ci, H=H, ubf=ubf))
            v = vn
        Wo, bo = p['expansion.outc.conv.weight'], p['expansion.outc.conv.bias']
        Cout = Wo.shape[0]
        ctx.state, ctx.p, ctx.fc_sv, ctx.ups, ctx.vlast, ctx.n = state, p, fc_sv, ups, v, n
        tgt = state.get('target')
        ctx.fused_loss = tgt is not None
        if tgt is not None:
            # training_step (lit.py:32-38): OutConv + MSE / MAE / MAPE + OutConv's data and weight gradient sums in one
            # pass; yhat and dL/dyhat are never written (mo_outc_loss_fwd)
            HW = v.H * v.W
            da = _empty(n, v.C, v.H, v.W, dev=dev, bf=bool(v.bf) and not GRAD_F32)
            ws = torch.empty(L.load().mo_outc_loss_ws_floats(n, HW, v.C, Cout), device=dev, dtype=torch.float32)
            out4 = torch.empty(4, device=dev, dtype=torch.float32)
            L.call('mo_outc_loss_fwd', L.ptr(v.t), v.istride, v.C, L.ptr(v.sc), L.ptr(v.sh), 1, gs, L.ptr(Wo), L.ptr(bo),
                   Cout, tgt.data_ptr(), L.ptr(state.get('target_off')), n, HW, None, L.ptr(da), v.C * HW, L.ptr(ws),
                   L.ptr(out4), (L.BF_IN0 * v.bf) | (L.BF_OUT * _is_bf(da)), st)
            ctx.loss_da, ctx.loss_ws = da, ws
            return out4
        out = _empty(n, Cout, v.H, v.W, dev=dev)
        L.call('mo_nchw_conv1x1_fwd', L.ptr(v.t), v.istride, v.C, L.ptr(v.sc), L.ptr(v.sh), 1, gs, L.ptr(Wo), L.ptr(bo),
               Cout, n, v.H * v.W, L.ptr(out), Cout * v.H * v.W, L.BF_IN0 * v.bf, st)
        return out

    @staticmethod
    def backward(ctx, dout):
        state, p, n, v = ctx.state, ctx.p, ctx.n, ctx.vlast
        lib = L.load()
        gs = state['gsize']
        dev = dout.device
        st = L.stream()
        grads = Grads(state.get('grad_out'), dev)
        dout = dout.contiguous()
        Wo = p['expansion.outc.conv.weight']
        Cout, C4 = Wo.shape[0], Wo.shape[1]
        HW = v.H * v.W
        dWo = grads.buf('expansion.outc.conv.weight', Wo.shape)
        dbo = grads.buf('expansion.outc.conv.bias', (Cout,))
        lane = _Lane(dev)
        if ctx.fused_loss:
            # dout = d / d(mse, mae, mape, rmse): its first element is the upstream gradient of the loss, a device scalar the
            # first activation backward and the OutConv weight-gradient reduction multiply by
            return UnetDecodeFn._backward_body(ctx, grads, lane, ctx.loss_da, dout, dWo, dbo)

        def outc_wgrad(ls):
            ws = lane.keep(torch.empty(max(lib.mo_unet_wgrad_ws_floats(Cout, C4, n * HW), n * Cout * 2), device=dev,
                                       dtype=torch.float32))
            L.call('mo_nchw_conv1x1_bwd_weight', L.ptr(dout), Cout * HW, Cout, L.ptr(v.t), v.istride, C4, L.ptr(v.sc),
                   L.ptr(v.sh), 1, gs, n, HW, L.ptr(dWo), L.ptr(dbo), L.ptr(ws), L.BF_IN0 * v.bf, ls)
        lane.run(outc_wgrad, reads=[dout, v.t, v.sc, v.sh, dWo, dbo])
        da = _empty(n, C4, v.H, v.W, dev=dev, bf=bool(v.bf) and not GRAD_F32)
        L.call('mo_nchw_conv1x1_bwd_data', L.ptr(dout), Cout * HW, Cout, L.ptr(Wo), C4, n, HW, L.ptr(da), C4 * HW,
               L.BF_OUT * _is_bf(da), st)
        return UnetDecodeFn._backward_body(ctx, grads, lane, da, None, None, None)

    @staticmethod
    def _backward_body(ctx, grads, lane, da, loss_scale, dWo, dbo):
        state, p, n, v = ctx.state, ctx.p, ctx.n, ctx.vlast
        lib = L.load()
        gs = state['gsize']
        dev = da.device
        st = L.stream()
        if loss_scale is not None:
            # the OutConv's weight / bias gradient: a second pass over the same view and target, beside the chain
            Cout, C4 = dWo.shape[0], dWo.shape[1]
            ws, tgt, toff = ctx.loss_ws, state['target'], state.get('target_off')
            Wo, bo = p['expansion.outc.conv.weight'], p['expansion.outc.conv.bias']
            lane.run(lambda ls: L.call('mo_outc_loss_bwd', L.ptr(v.t), v.istride, v.C, L.ptr(v.sc), L.ptr(v.sh), 1, gs,
                                       L.ptr(Wo), L.ptr(bo), Cout, tgt.data_ptr(), L.ptr(toff), n, v.H * v.W, L.ptr(ws),
                                       L.ptr(loss_scale), L.ptr(dWo), L.ptr(dbo), L.BF_IN0 * v.bf, ls),
                     reads=[ws, loss_scale, dWo, dbo, v.t, v.sc, v.sh, tgt, toff])
        dfm = [None] * 4
        for k in (4, 3, 2, 1):
            up = ctx.ups[k - 1]
            ci, H, vin = up['ci'], up['H'], up['vin']
            dcat = double_conv_bwd(p, up['dc'], n, gs, grads, dev, da=da, dp=None, lane=lane, dx_bf=up['ubf'],
                                   da_scale=loss_scale if k == 4 else None)     # (n, ci, 2H, 2H)
            dyf = L.BF_DY * _is_bf(dcat)
            C0 = ci // 2
            dfm[4 - k] = dcat[:, :C0]
            du = dcat[:, C0:]
            du_stride = dcat.stride(0)
            Wt = p[f'expansion.up{k}.up.weight']
            dWt = grads.buf(f'expansion.up{k}.up.weight', Wt.shape)
            dbt = grads.buf(f'expansion.up{k}.up.bias', (C0,))

            def convt_wgrad(ls, du=du, du_stride=du_stride, C0=C0, vin=vin, ci=ci, H=H, dWt=dWt, dbt=dbt, dyf=dyf):
                wsu = lane.keep(torch.empty(max(lib.mo_unet_wgrad_ws_floats(ci, 4 * C0, n * H * H), n * C0 * 2), device=dev,
                                            dtype=torch.float32))
                L.call('mo_convt2x2_bwd_weight', du.data_ptr(), du_stride, C0, L.ptr(vin.t), vin.istride, ci,
                       L.ptr(vin.sc), L.ptr(vin.sh), 1 if vin.sc is not None else 0, gs, n, H, H, L.ptr(dWt), L.ptr(dbt),
                       L.ptr(wsu), dyf | (L.BF_IN0 * vin.bf), ls)
            lane.run(convt_wgrad, reads=[dcat, vin.t, vin.sc, vin.sh, dWt, dbt])
            da = _empty(n, ci, H, H, dev=dev)
            L.call('mo_convt2x2_bwd_data', du.data_ptr(), du_stride, C0, L.ptr(Wt), ci, n, H, H, L.ptr(da), ci * H * H,
                   dyf, st)
        dz = fc_block_bwd(p, ctx.fc_sv, da.view(n, -1), grads, lane=lane)
        if state.get('adam_now') is not None and state.get('grad_out'):
            lane.run(lambda ls: state['adam_now'](('decoder.', 'expansion.'), ls))      # (eager Adam, see UnetEncodeFn.backward)
        gout = state.get('grad_out') or {}
        if state.get('defer_join') and all(gout.get(k) is not None for k in state['names']):
            lane.defer()                         # (every gradient of this pass went into the trainer's buffers in place)
        else:
            lane.join()
        ctx.fc_sv = ctx.ups = ctx.vlast = ctx.loss_da = ctx.loss_ws = None      # (release the activations, see UnetEncodeFn)
        if state.get('skip_side') is not None:
            state['skip_side']['dfm'] = dfm      # (picked up by UnetEncodeFn.backward)
            return (None, dz, None, None, None, None) + grads.result(state['names'])
        return (None, dz, dfm[0], dfm[1], dfm[2], dfm[3]) + grads.result(state['names'])
