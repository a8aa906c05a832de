"""GPU parity tests for the UNet C-ABI ops (unet.py:40-92) against plain fp32 torch on CPU; 1e-4."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import rand
from test_ops_gpu import dev, close, _release, L  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


def act_view(y, sc, sh, gsize):
    """relu(y*sc+sh) with per-(group,channel) affine; y (n,C,H,W), sc/sh (G,C)."""
    n, C = y.shape[:2]
    s = sc.repeat_interleave(gsize, 0).view(n, C, 1, 1)
    t = sh.repeat_interleave(gsize, 0).view(n, C, 1, 1)
    return F.relu(y * s + t)


@pytest.mark.parametrize('n,gs,C0,C1,Co,H,W', [(6, 3, 1, 0, 4, 16, 16), (4, 2, 8, 0, 8, 8, 12), (4, 2, 16, 16, 16, 8, 8),
                                                (2, 1, 64, 0, 64, 8, 8), (3, 3, 4, 4, 4, 32, 32), (2, 2, 32, 32, 32, 16, 16),
                                                # direct-convolution path (thin layers at >= 32x32): ragged tiles, odd channel counts
                                                (2, 1, 13, 0, 4, 32, 64), (2, 2, 5, 3, 13, 48, 80), (1, 1, 20, 12, 32, 32, 36),
                                                (2, 1, 4, 0, 8, 128, 128),
                                                # MFMA weight-gradient path (W % 64 == 0, H % 8 == 0): 3 / 5 / 9 m-blocks, two views,
                                                # two input-channel chunks (ragged second chunk), 32 output channels, many images per workgroup
                                                (4, 2, 13, 0, 4, 64, 64), (3, 1, 8, 0, 8, 16, 128), (2, 2, 4, 4, 4, 24, 64), (2, 1, 16, 16, 16, 64, 64),
                                                (2, 2, 12, 8, 32, 32, 64), (40, 2, 4, 0, 4, 64, 128), (2, 1, 5, 0, 16, 8, 64),
                                                # deep levels on the matrix pipe (ux_conv3x3_mfma_kernel): 1..4 blocks of 16 output channels,
                                                # 8x32 and 16x16 tiles, two views, ragged channel chunks
                                                (2, 1, 16, 0, 32, 32, 32), (2, 2, 32, 0, 64, 16, 16), (2, 1, 32, 32, 32, 32, 32),
                                                (3, 1, 24, 8, 48, 16, 32), (2, 2, 8, 0, 20, 64, 64), (2, 1, 13, 0, 8, 16, 16),
                                                # MFMA weight gradient on 16x32 and 32x16 tiles
                                                (3, 1, 16, 0, 32, 32, 32), (2, 2, 32, 32, 32, 48, 32), (2, 1, 8, 0, 16, 64, 16), (5, 1, 20, 0, 12, 32, 48)])
@pytest.mark.parametrize('bf', [0, 1])
def test_conv3x3_fwd_bwd(L, n, gs, C0, C1, Co, H, W, bf):
    """bf=1: the same layer with its activation tensors (in0, out, dy; in1 = the fp32 upsampled map of an Up block)
    stored as bf16 -- only on shapes the streaming kernels serve (include/mo_hip.h `dtypes`)."""
    lib = L.load()
    if bf and not (Co <= 32 and H % 8 == 0 and W % 64 == 0 and H >= 32):
        pytest.skip('bf16 storage exists on the direct / MFMA kernels only')
    G = n // gs
    q = (lambda t: t.to(torch.bfloat16).float()) if bf else (lambda t: t)
    x0 = q(rand(1, (n, C0, H, W))).requires_grad_(True)
    sc0, sh0 = rand(2, (G, C0)) * 0.3 + 1, rand(3, (G, C0)) * 0.3
    a0 = act_view(x0, sc0, sh0, gs)
    ins = [a0]
    if C1:
        x1 = rand(4, (n, C1, H, W)).requires_grad_(True)
        ins.append(x1)
    Wt = (rand(5, (Co, C0 + C1, 3, 3)) / np.sqrt(9 * (C0 + C1))).requires_grad_(True)
    ref = F.conv2d(torch.cat(ins, 1), Wt, None, padding=1)
    out = torch.empty(n, Co, H, W, device='cuda', dtype=torch.bfloat16 if bf else torch.float32)
    x0d = dev(x0.detach().to(torch.bfloat16) if bf else x0.detach())
    x1d = dev(x1.detach()) if C1 else None
    args_in = (L.ptr(x0d), C0, C0 * H * W, L.ptr(dev(sc0)), L.ptr(dev(sh0)), 1,
               L.ptr(x1d), C1, C1 * H * W, None, None, 0)
    ntile = lib.mo_conv3x3_stats_tiles(Co, n, H, W)
    stats = torch.full((n, max(ntile, 1), Co, 2), float('nan'), device='cuda')
    L.call('mo_conv3x3_fwd', *args_in, gs, L.ptr(dev(Wt.detach())), Co, n, H, W, L.ptr(out), Co * H * W,
           L.ptr(stats) if ntile else None, (L.BF_IN0 | L.BF_OUT) if bf else 0, None, L.stream())
    close(out.float(), ref, tol=4e-3 if bf else 1e-4, what='conv fwd')          # bf16 result: half an ulp = 2^-9
    if ntile:
        # BatchNorm statistics from the conv epilogue: per-tile (sum, sum of squares) rows add up to the per-image sums
        st = stats.cpu().double().sum(1)
        rd = ref.detach().double()
        close(st[..., 0], rd.sum((2, 3)), tol=1e-5, what='epilogue sum')
        close(st[..., 1], (rd * rd).sum((2, 3)), tol=1e-5, what='epilogue sumsq')
    dy = q(rand(6, tuple(ref.shape)))
    ref.backward(dy)
    # data gradient = conv of dy with the flipped/transposed weights -> gradient w.r.t. the *activated* cat
    Ci = C0 + C1
    Wf = torch.empty(Ci, Co, 3, 3, device='cuda')
    L.call('mo_conv3x3_flip_weights', L.ptr(dev(Wt.detach())), Co, Ci, L.ptr(Wf), L.stream())
    dyd = dev(dy.to(torch.bfloat16) if bf else dy)
    for obf in ((0, 1) if bf else (0,)):                   # data gradient into an fp32 tensor (Up blocks) or a bf16 one
        dcat = torch.empty(n, Ci, H, W, device='cuda', dtype=torch.bfloat16 if obf else torch.float32)
        L.call('mo_conv3x3_fwd', L.ptr(dyd), Co, Co * H * W, None, None, 0, None, 0, 0, None, None, 0, 1, L.ptr(Wf), Ci,
               n, H, W, L.ptr(dcat), Ci * H * W, None, (L.BF_IN0 * bf) | (L.BF_OUT * obf), None, L.stream())
        dcat_ref = F.conv_transpose2d(dy, Wt.detach(), padding=1)
        close(dcat.float(), dcat_ref, tol=4e-3 if obf else 1e-4, what='conv bwd data')
        if C1:
            close(dcat[:, C0:].float(), x1.grad, tol=4e-3 if obf else 1e-4, what='dx1')
    dW = torch.empty(Co, Ci, 3, 3, device='cuda')
    ws = torch.empty(lib.mo_unet_wgrad_ws_floats(Co, Ci * 9, n * H * W), device='cuda')
    L.call('mo_conv3x3_bwd_weight', L.ptr(dyd), Co * H * W, Co, *args_in, gs, n, H, W, L.ptr(dW), L.ptr(ws),
           (L.BF_DY | L.BF_IN0) if bf else 0, None, L.stream())
    close(dW, Wt.grad, what='conv dW')


@pytest.mark.parametrize('n,C0,Co,H,W', [(6, 1, 4, 16, 16), (2, 13, 8, 16, 16), (3, 20, 32, 32, 32), (2, 5, 16, 32, 32)])
def test_conv3x3_partial_channel_chunk_ignores_stale_lds(L, n, C0, Co, H, W):
    """The deep-level conv stages 8 input channels at a time; in a last chunk of fewer than 8 the lanes past its last k
    multiply by zero -- and must not read what an EARLIER kernel left in LDS (0 * NaN = NaN: an order-dependent failure of
    the suite in round 3).  A launch on NaN data of 8 channels leaves NaN patterns in every staged LDS word; the ragged
    launch right behind it on the same CUs must still be finite and right."""
    lib = L.load()
    nb = 1200 * 256 // (H * W) + n                        # (enough workgroups to visit every CU several times)
    nanx = torch.full((nb, 8, H, W), float('nan'), device='cuda')
    nanw = torch.full((Co, 8, 3, 3), float('nan'), device='cuda')
    junk = torch.empty(nb, Co, H, W, device='cuda')
    x = rand(91, (n, C0, H, W)); Wt = rand(92, (Co, C0, 3, 3)) / np.sqrt(9 * C0)
    xd, Wd_ = dev(x), dev(Wt)
    out = torch.empty(n, Co, H, W, device='cuda')
    for _ in range(3):                                        # (several rounds: every CU sees both launches)
        L.call('mo_conv3x3_fwd', L.ptr(nanx), 8, 8 * H * W, None, None, 0, None, 0, 0, None, None, 0, 1, L.ptr(nanw), Co, nb, H, W,
               L.ptr(junk), Co * H * W, None, 0, None, L.stream())
        L.call('mo_conv3x3_fwd', L.ptr(xd), C0, C0 * H * W, None, None, 0, None, 0, 0, None, None, 0, 1, L.ptr(Wd_), Co, n, H, W,
               L.ptr(out), Co * H * W, None, 0, None, L.stream())
        assert bool(torch.isfinite(out).all())
        close(out, F.conv2d(x, Wt, None, padding=1), tol=1e-4, what='conv fwd behind a NaN launch')


@pytest.mark.parametrize('n,C0,Co,H,W', [(3, 24, 64, 16, 16), (2, 32, 64, 32, 32), (5, 12, 32, 16, 16), (2, 16, 48, 16, 32),
                                        (2, 8, 40, 32, 32)])
def test_conv3x3_output_channels_dealt_over_workgroups(L, n, C0, Co, H, W):
    """The deep-level conv deals the 16-channel output blocks of a tile over 1, 2 or 4 workgroups when the grid would not
    fill the chip (option ux_split; 0 = the heuristic).  Every split computes each output element with the same operands
    in the same order: outputs and BatchNorm statistics rows are bit-identical to the unsplit launch (a split the block
    count does not divide falls back to it)."""
    lib = L.load()
    x = rand(71, (n, C0, H, W)); Wt = rand(72, (Co, C0, 3, 3)) / np.sqrt(9 * C0)
    xd, Wd_ = dev(x), dev(Wt)
    ntile = lib.mo_conv3x3_stats_tiles(Co, n, H, W)
    assert ntile > 0
    res = {}
    try:
        for split in (1, 2, 4, 0):
            L.call('mo_unet_set_option', b'ux_split', split)
            out = torch.full((n, Co, H, W), float('nan'), device='cuda')
            stats = torch.full((n, ntile, Co, 2), float('nan'), device='cuda')
            L.call('mo_conv3x3_fwd', L.ptr(xd), C0, C0 * H * W, None, None, 0, None, 0, 0, None, None, 0, 1, L.ptr(Wd_), Co, n, H, W,
                   L.ptr(out), Co * H * W, L.ptr(stats), 0, None, L.stream())
            res[split] = (out.cpu(), stats.cpu())
    finally:
        L.call('mo_unet_set_option', b'ux_split', 0)
    close(res[1][0], F.conv2d(x, Wt, None, padding=1), tol=1e-4, what='conv fwd, one workgroup per tile')
    for split in (2, 4, 0):
        assert torch.equal(res[split][0], res[1][0]), f'outputs differ at ux_split={split}'
        assert torch.equal(res[split][1], res[1][1]), f'statistics rows differ at ux_split={split}'


@pytest.mark.parametrize('n,gs,C0,C1,Co,H,W', [(4, 2, 4, 0, 4, 16, 64), (3, 1, 13, 0, 4, 32, 64), (2, 2, 4, 4, 4, 16, 128),
                                                (2, 1, 8, 0, 8, 32, 128), (2, 2, 8, 8, 8, 16, 64), (2, 1, 8, 0, 16, 64, 64),
                                                (2, 1, 16, 16, 16, 16, 64), (3, 3, 16, 0, 32, 16, 64), (2, 1, 16, 16, 32, 32, 64), (2, 1, 20, 0, 32, 32, 64),
                                                (2, 2, 5, 0, 13, 16, 64), (70, 2, 4, 0, 8, 64, 128), (40, 2, 8, 8, 20, 16, 64),
                                                # 32-pixel-wide tiles (the 32 x 32 level)
                                                (3, 1, 16, 0, 32, 32, 32), (2, 2, 32, 0, 32, 32, 32), (2, 1, 32, 0, 16, 16, 32),
                                                (4, 2, 8, 8, 8, 32, 32), (2, 1, 4, 0, 4, 48, 32)])
@pytest.mark.parametrize('bf', [0, 1])
def test_conv3x3_bf16_matrix_pipe(L, n, gs, C0, C1, Co, H, W, bf):
    """MO_BF_MATH: the 3x3 conv, its data gradient (weights read transposed + flipped, MO_W_FLIP) and its weight gradient
    on the bf16 matrix pipe (unet_bf16.hpp).  Operands enter the MFMA rounded to bf16 and every sum is fp32, so the
    yardstick is the fp32 CPU conv of the SAME bf16-rounded operands (1e-3 of scale: accumulation order, plus the few
    activations whose fused-multiply-add affine lands on the other side of a bf16 rounding boundary); the unrounded conv
    is checked at the bf16 tolerance.  bf=1: activations stored as bf16 as well."""
    lib = L.load()
    if W % 64:
        L.call('mo_unet_set_option', b'ub_min_w', 32)       # (the 32-pixel-wide tiles are off by default)
    try:
        _conv3x3_bf16_matrix_pipe(L, lib, n, gs, C0, C1, Co, H, W, bf)
    finally:
        L.call('mo_unet_set_option', b'ub_min_w', 64)


def _conv3x3_bf16_matrix_pipe(L, lib, n, gs, C0, C1, Co, H, W, bf):
    assert lib.mo_conv3x3_bf16_route(C0 + C1, Co, n, H, W) == 1 and (C1 == 0 or C0 == C1)
    G = n // gs
    rb = lambda t: t.to(torch.bfloat16).float()
    q = rb if bf else (lambda t: t)
    x0 = q(rand(1, (n, C0, H, W)))
    sc0, sh0 = rand(2, (G, C0)) * 0.3 + 1, rand(3, (G, C0)) * 0.3
    ins = [act_view(x0, sc0, sh0, gs)]
    if C1:
        x1 = rand(4, (n, C1, H, W))
        ins.append(x1)
    Ci = C0 + C1
    Wt = rand(5, (Co, Ci, 3, 3)) / np.sqrt(9 * Ci)
    cat = torch.cat(ins, 1)
    ref_q = F.conv2d(rb(cat), rb(Wt), None, padding=1)
    ref = F.conv2d(cat, Wt, None, padding=1)
    out = torch.empty(n, Co, H, W, device='cuda', dtype=torch.bfloat16 if bf else torch.float32)
    x0d = dev(x0.to(torch.bfloat16) if bf else x0)
    x1d = dev(x1) if C1 else None
    args_in = (L.ptr(x0d), C0, C0 * H * W, L.ptr(dev(sc0)), L.ptr(dev(sh0)), 1, L.ptr(x1d), C1, C1 * H * W, None, None, 0)
    dt = L.BF_MATH | ((L.BF_IN0 | L.BF_OUT) if bf else 0)
    ntile = lib.mo_conv3x3_stats_tiles2(C0, C1, Co, n, H, W, dt)
    assert ntile == H // 16                    # one statistics row per image and 16-row band
    stats = torch.full((n, ntile, Co, 2), float('nan'), device='cuda')
    Wd_ = dev(Wt)
    L.call('mo_conv3x3_fwd', *args_in, gs, L.ptr(Wd_), Co, n, H, W, L.ptr(out), Co * H * W, L.ptr(stats), dt, None, L.stream())
    close(out.float(), ref_q, tol=4e-3 if bf else 1e-3, what='conv fwd vs bf16-rounded operands')
    close(out.float(), ref, tol=1.5e-2, what='conv fwd vs fp32 conv')
    st = stats.cpu().double().sum(1)
    rd = ref_q.double()
    close(st[..., 0], rd.sum((2, 3)), tol=1e-3, what='epilogue sum')
    close(st[..., 1], (rd * rd).sum((2, 3)), tol=1e-3, what='epilogue sumsq')
    # data gradient: the same kernel on dy, weights read transposed + flipped from the forward tensor
    dy = q(rand(6, (n, Co, H, W)))
    dyd = dev(dy.to(torch.bfloat16) if bf else dy)
    dref_q = F.conv_transpose2d(rb(dy), rb(Wt), padding=1)
    for obf in ((0, 1) if bf else (0,)):
        dcat = torch.empty(n, Ci, H, W, device='cuda', dtype=torch.bfloat16 if obf else torch.float32)
        L.call('mo_conv3x3_fwd', L.ptr(dyd), Co, Co * H * W, None, None, 0, None, 0, 0, None, None, 0, 1, L.ptr(Wd_), Ci,
               n, H, W, L.ptr(dcat), Ci * H * W, None, L.BF_MATH | L.W_FLIP | (L.BF_IN0 * bf) | (L.BF_OUT * obf), None, L.stream())
        close(dcat.float(), dref_q, tol=4e-3 if obf else 1e-4, what='conv bwd data vs bf16-rounded operands')
    if bf and Co <= 16 and W % 64 == 0:
        # weight gradient: dy read straight into the B fragments, the activated input planar in LDS
        Wp = Wt.clone().requires_grad_(True)
        F.conv2d(rb(cat), Wp, None, padding=1).backward(rb(dy))
        dW = torch.full((Co, Ci, 3, 3), float('nan'), device='cuda')
        ws = torch.empty(lib.mo_unet_wgrad_ws_floats(Co, Ci * 9, n * H * W), device='cuda')
        L.call('mo_conv3x3_bwd_weight', L.ptr(dyd), Co * H * W, Co, *args_in, gs, n, H, W, L.ptr(dW), L.ptr(ws),
               L.BF_MATH | L.BF_DY | L.BF_IN0, None, L.stream())
        close(dW, Wp.grad, tol=1e-3, what='conv dW vs bf16-rounded operands')


@pytest.mark.parametrize('n,gs,C,H,W,pool,use_da', [(6, 3, 4, 16, 16, True, True), (4, 2, 8, 8, 8, True, False),
                                                    (4, 1, 16, 8, 12, False, True), (2, 2, 64, 8, 8, False, True)])
@pytest.mark.parametrize('bf', [0, 1])
def test_group_bn_act_pool_fwd_bwd(L, n, gs, C, H, W, pool, use_da, bf):
    """[conv out] -> BatchNorm2d (train, per group of gs images) -> ReLU (-> MaxPool2d(2)) and backward.
    bf=1: y, da, dp, the pooled output and dy stored as bf16."""
    lib = L.load()
    G = n // gs
    q = (lambda t: t.to(torch.bfloat16).float()) if bf else (lambda t: t)
    hb = (lambda t: t.to(torch.bfloat16)) if bf else (lambda t: t)
    y = q(rand(10, (n, C, H, W))).requires_grad_(True)
    gamma = (rand(11, (C,)) * 0.3 + 1).requires_grad_(True)
    beta = (rand(12, (C,)) * 0.3).requires_grad_(True)
    rm0, rv0 = rand(13, (C,)) * 0.1, rand(14, (C,)).abs() + 0.5
    rm, rv = rm0.clone(), rv0.clone()
    acts = []
    for g in range(G):                                   # reference: one BN call per group, in order
        acts.append(F.relu(F.batch_norm(y[g * gs:(g + 1) * gs], rm, rv, gamma, beta, True, 0.1, 1e-5)))
    a = torch.cat(acts, 0)
    yd = dev(hb(y.detach()))
    stats = torch.empty(n, C, 2, device='cuda')
    L.call('mo_nchw_stats', L.ptr(dev(y.detach())), C * H * W, C, n, H * W, L.ptr(stats), L.stream())
    rmd, rvd = dev(rm0.clone()), dev(rv0.clone())
    aff = torch.empty(4, G, C, device='cuda')
    gd, bd = dev(gamma.detach()), dev(beta.detach())
    L.call('mo_group_bn_finalize', L.ptr(stats), n, C, gs, H * W, 1, L.ptr(gd), L.ptr(bd), L.ptr(rmd), L.ptr(rvd), 0.1, 1e-5, 1,
           L.ptr(aff[0]), L.ptr(aff[1]), L.ptr(aff[2]), L.ptr(aff[3]), L.stream())
    close(rmd, rm, 1e-5, 'running_mean (sequential group updates)')
    close(rvd, rv, 1e-5, 'running_var')
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    out = torch.empty(n, C, Ho, Wo, device='cuda', dtype=torch.bfloat16 if bf else torch.float32)
    L.call('mo_unet_act', L.ptr(yd), C * H * W, C, n, H, W, L.ptr(aff[0]), L.ptr(aff[1]), gs, 1 if pool else 0,
           L.ptr(out), C * Ho * Wo, (L.BF_IN0 | L.BF_OUT) if bf else 0, L.stream())
    ref_out = F.max_pool2d(a, 2) if pool else a
    close(out.float(), ref_out, tol=4e-3 if bf else 1e-4, what='act/pool')
    # backward: loss = <a, da> + <pool(a), dp>
    da = q(rand(15, (n, C, H, W))) if use_da else None
    dp = q(rand(16, (n, C, H // 2, W // 2))) if pool else None
    loss = 0
    if use_da:
        loss = loss + (a * da).sum()
    if pool:
        loss = loss + (F.max_pool2d(a, 2) * dp).sum()
    loss.backward()
    dy = torch.empty(n, C, H, W, device='cuda', dtype=torch.bfloat16 if bf else torch.float32)
    dgam = torch.empty(C, device='cuda'); dbet = torch.empty(C, device='cuda')
    ws = torch.empty(lib.mo_unet_act_bwd_ws_floats(n, C), device='cuda')
    L.call('mo_unet_act_bwd', L.ptr(yd), C * H * W, C, n, H, W, gs, L.ptr(gd), L.ptr(aff[2]), L.ptr(aff[3]),
           L.ptr(aff[0]), L.ptr(aff[1]), L.ptr(dev(hb(da))) if use_da else None, C * H * W,
           L.ptr(dev(hb(dp))) if pool else None, C * (H // 2) * (W // 2), L.ptr(dy), C * H * W, L.ptr(dgam), L.ptr(dbet),
           L.ptr(ws), (L.BF_IN0 | L.BF_IN1 | L.BF_DP | L.BF_OUT) if bf else 0, None, L.stream())
    close(dy.float(), y.grad, tol=4e-3 if bf else 1e-4, what='dy')
    close(dgam, gamma.grad, what='dgamma')
    close(dbet, beta.grad, what='dbeta')


@pytest.mark.parametrize('n,gs,Ci,Co,H,W,act', [(4, 2, 64, 32, 8, 8, False), (4, 2, 8, 4, 16, 16, True), (2, 1, 16, 8, 4, 8, True),
                                                # streaming kernels of the thin Up blocks: ragged channel chunks, many images per workgroup
                                                (6, 2, 8, 4, 64, 64, True), (3, 1, 16, 8, 32, 48, True), (2, 2, 6, 3, 8, 20, False),
                                                (70, 2, 8, 4, 32, 32, True), (2, 1, 10, 5, 16, 16, True)])
def test_convt2x2_fwd_bwd(L, n, gs, Ci, Co, H, W, act):
    lib = L.load()
    G = n // gs
    x = rand(20, (n, Ci, H, W)).requires_grad_(True)
    sc, sh = rand(21, (G, Ci)) * 0.3 + 1, rand(22, (G, Ci)) * 0.3
    a = act_view(x, sc, sh, gs) if act else x
    a.retain_grad()
    Wt = (rand(23, (Ci, Co, 2, 2)) / np.sqrt(Ci)).requires_grad_(True)
    b = rand(24, (Co,)).requires_grad_(True)
    ref = F.conv_transpose2d(a, Wt, b, stride=2)
    xd = dev(x.detach())
    scd, shd = (dev(sc), dev(sh)) if act else (None, None)
    out = torch.empty(n, Co, 2 * H, 2 * W, device='cuda')
    L.call('mo_convt2x2_fwd', L.ptr(xd), Ci * H * W, Ci, L.ptr(scd), L.ptr(shd), 1 if act else 0, gs,
           L.ptr(dev(Wt.detach())), L.ptr(dev(b.detach())), Co, n, H, W, L.ptr(out), Co * 4 * H * W, 0, L.stream())
    close(out, ref, what='convT fwd')
    dout = rand(25, tuple(ref.shape))
    ref.backward(dout)
    dd = dev(dout)
    din = torch.empty(n, Ci, H, W, device='cuda')
    L.call('mo_convt2x2_bwd_data', L.ptr(dd), Co * 4 * H * W, Co, L.ptr(dev(Wt.detach())), Ci, n, H, W, L.ptr(din),
           Ci * H * W, 0, L.stream())
    close(din, a.grad, what='convT bwd data')
    dW = torch.empty(Ci, Co, 2, 2, device='cuda')
    db = torch.empty(Co, device='cuda')
    ws = torch.empty(max(lib.mo_unet_wgrad_ws_floats(Ci, 4 * Co, n * H * W), n * Co * 2), device='cuda')
    L.call('mo_convt2x2_bwd_weight', L.ptr(dd), Co * 4 * H * W, Co, L.ptr(xd), Ci * H * W, Ci, L.ptr(scd), L.ptr(shd),
           1 if act else 0, gs, n, H, W, L.ptr(dW), L.ptr(db), L.ptr(ws), 0, L.stream())
    close(dW, Wt.grad, what='convT dW')
    close(db, b.grad, what='convT db')
    db_ref = b.grad.clone()
    # bf16 storage of the upsampled map and of the gradient w.r.t. it (the streaming kernels only): the same fp32
    # arithmetic on the rounded tensors -- result rounded once, gradients exact in the bf16-rounded dout
    if lib.mo_convt2x2_bf16_route(Ci, Co, n) and W % 4 == 0:
        outb = torch.empty(n, Co, 2 * H, 2 * W, device='cuda', dtype=torch.bfloat16)
        L.call('mo_convt2x2_fwd', L.ptr(xd), Ci * H * W, Ci, L.ptr(scd), L.ptr(shd), 1 if act else 0, gs,
               L.ptr(dev(Wt.detach())), L.ptr(dev(b.detach())), Co, n, H, W, L.ptr(outb), Co * 4 * H * W, L.BF_OUT, L.stream())
        assert torch.equal(outb.cpu(), out.cpu().to(torch.bfloat16))
        doutb = dout.to(torch.bfloat16)
        x2 = x.detach().clone().requires_grad_(True)
        a = act_view(x2, sc, sh, gs) if act else x2
        a.retain_grad()
        Wt.grad = None; b.grad = None
        F.conv_transpose2d(a, Wt, b, stride=2).backward(doutb.float())
        ddb = dev(doutb)
        L.call('mo_convt2x2_bwd_data', L.ptr(ddb), Co * 4 * H * W, Co, L.ptr(dev(Wt.detach())), Ci, n, H, W, L.ptr(din),
               Ci * H * W, L.BF_DY, L.stream())
        close(din, a.grad, what='convT bwd data, bf16 dout')
        L.call('mo_convt2x2_bwd_weight', L.ptr(ddb), Co * 4 * H * W, Co, L.ptr(xd), Ci * H * W, Ci, L.ptr(scd), L.ptr(shd),
               1 if act else 0, gs, n, H, W, L.ptr(dW), L.ptr(db), L.ptr(ws), L.BF_DY, L.stream())
        close(dW, Wt.grad, what='convT dW, bf16 dout')
        close(db, b.grad, what='convT db, bf16 dout')
    db2 = torch.empty(Co, device='cuda')
    ws2 = torch.empty(n * Co * 2, device='cuda')
    L.call('mo_nchw_channel_sum', L.ptr(dd), Co * 4 * H * W, Co, n, 4 * H * W, L.ptr(db2), L.ptr(ws2), L.stream())
    close(db2, db_ref, what='channel sum')


@pytest.mark.parametrize('n,gs,Ci,Co,HW', [(4, 2, 4, 1, 256), (2, 1, 4, 3, 64), (6, 2, 4, 13, 64 * 64), (4, 2, 8, 8, 1028),
                                          (2, 2, 13, 4, 2048), (2, 1, 24, 20, 256), (3, 3, 4, 16, 5000)])
@pytest.mark.parametrize('bf', [0, 1])
def test_outconv_fwd_bwd(L, n, gs, Ci, Co, HW, bf):
    """bf=1: the input view (the last DoubleConv's raw output) and the data gradient stored as bf16."""
    lib = L.load()
    if bf and (Ci > 16 or Co > 16):
        pytest.skip('bf16 storage exists on the streaming kernels only')
    G = n // gs
    x = rand(30, (n, Ci, HW, 1))
    x = (x.to(torch.bfloat16).float() if bf else x).requires_grad_(True)
    sc, sh = rand(31, (G, Ci)) * 0.3 + 1, rand(32, (G, Ci)) * 0.3
    a = act_view(x, sc, sh, gs)
    a.retain_grad()
    Wt = rand(33, (Co, Ci, 1, 1)).requires_grad_(True)
    b = rand(34, (Co,)).requires_grad_(True)
    ref = F.conv2d(a, Wt, b)
    xd, scd, shd = dev(x.detach().to(torch.bfloat16) if bf else x.detach()), dev(sc), dev(sh)
    out = torch.empty(n, Co, HW, device='cuda')
    L.call('mo_nchw_conv1x1_fwd', L.ptr(xd), Ci * HW, Ci, L.ptr(scd), L.ptr(shd), 1, gs, L.ptr(dev(Wt.detach())),
           L.ptr(dev(b.detach())), Co, n, HW, L.ptr(out), Co * HW, L.BF_IN0 * bf, L.stream())
    close(out, ref.reshape(n, Co, HW), what='outc fwd')
    dout = rand(35, tuple(ref.shape))
    ref.backward(dout)
    dd = dev(dout)
    din = torch.empty(n, Ci, HW, device='cuda', dtype=torch.bfloat16 if bf else torch.float32)
    L.call('mo_nchw_conv1x1_bwd_data', L.ptr(dd), Co * HW, Co, L.ptr(dev(Wt.detach())), Ci, n, HW, L.ptr(din), Ci * HW,
           L.BF_OUT * bf, L.stream())
    close(din.float(), a.grad.reshape(n, Ci, HW), tol=4e-3 if bf else 1e-4, what='outc bwd data')
    dW = torch.empty(Co, Ci, device='cuda')
    db = torch.empty(Co, device='cuda')
    ws = torch.empty(max(lib.mo_unet_wgrad_ws_floats(Co, Ci, n * HW), n * Co * 2), device='cuda')
    L.call('mo_nchw_conv1x1_bwd_weight', L.ptr(dd), Co * HW, Co, L.ptr(xd), Ci * HW, Ci, L.ptr(scd), L.ptr(shd), 1, gs, n,
           HW, L.ptr(dW), L.ptr(db), L.ptr(ws), L.BF_IN0 * bf, L.stream())
    close(dW, Wt.grad.reshape(Co, Ci), what='outc dW')
    close(db, b.grad, what='outc db')


def test_dropout_mask(L):
    n = 100003
    x = rand(40, (n,))
    y = torch.empty(n, device='cuda')
    thresh = int(0.3 * 4294967296.0)
    L.call('mo_dropout', L.ptr(dev(x)), L.ptr(y), n, 77, thresh, 1.0 / 0.7, L.stream())
    yc = y.cpu()
    kept = yc != 0
    assert abs(float(kept.float().mean()) - 0.7) < 0.01
    assert torch.allclose(yc[kept], x[kept] / 0.7, rtol=1e-6)
    y2 = torch.empty(n, device='cuda')
    L.call('mo_dropout', L.ptr(dev(x)), L.ptr(y2), n, 77, thresh, 1.0 / 0.7, L.stream())
    assert torch.equal(y, y2)


@pytest.mark.parametrize('P,K,N', [(134, 16384, 4096), (134, 1024, 16384), (7, 256, 128), (144, 4096, 264), (33, 520, 72),
                                   # several windows per step: row groups of 144 walked inside the kernel (ragged last group)
                                   (536, 16384, 4096), (300, 1024, 520), (145, 256, 128),
                                   # ragged shapes on the two-column-block kernel (>= 2048 columns, >= 4096 reduction)
                                   (150, 4104, 2056), (20, 2056, 4104)])
def test_fc_three_way_bf16_split(L, P, K, N):
    """mo_fc3_fwd / mo_fc3_bwd_data (csrc/unet_fc.hpp): few-row Linear layers with "3 x bf16" split products on the bf16
    matrix pipe against the float64 product: 5e-5 of the result's scale (a plain bf16 product would sit at ~2e-3);
    ragged row counts, reduction lengths that are not a multiple of the 128- / 64-element chunk, column counts that are not
    a multiple of a workgroup's 64 / 128."""
    lib = L.load()
    x = rand(31, (P, K)); W = rand(32, (N, K)) / np.sqrt(K); b = rand(33, (N,)); dout = rand(34, (P, N))
    xd, Wd, bd, dd = dev(x), dev(W), dev(b), dev(dout)
    assert lib.mo_fc3_supported(P, K, N) == 1 and lib.mo_fc3_supported(P, N, K) == 1
    for relu in (0, 1):
        out = torch.full((P, N), float('nan'), device='cuda')
        ws = torch.empty(lib.mo_fc3_ws_floats(P, K, N), device='cuda')
        L.call('mo_fc3_fwd', L.ptr(xd), P, K, L.ptr(Wd), L.ptr(bd), N, relu, L.ptr(out), L.ptr(ws), L.stream())
        ref = x.double() @ W.double().t() + b.double()
        close(out, torch.relu(ref) if relu else ref, tol=5e-5, what='fc3 forward')
    din = torch.full((P, K), float('nan'), device='cuda')
    ws = torch.empty(lib.mo_fc3_ws_floats(P, N, K), device='cuda')
    L.call('mo_fc3_bwd_data', L.ptr(dd), P, N, L.ptr(Wd), K, L.ptr(din), L.ptr(ws), L.stream())
    close(din, dout.double() @ W.double(), tol=5e-5, what='fc3 data gradient')
    if P > 160:
        return                                   # (the weight-gradient kernel serves one window: P <= 160 rows)
    dW = torch.full((N, K), float('nan'), device='cuda'); db = torch.full((N,), float('nan'), device='cuda')
    ws = torch.empty(lib.mo_fc3_wgrad_ws_floats(P, N, K), device='cuda')
    L.call('mo_fc3_bwd_weight', L.ptr(dd), P, N, L.ptr(xd), K, L.ptr(dW), L.ptr(db), L.ptr(ws), L.stream())
    close(dW, dout.double().t() @ x.double(), tol=5e-5, what='fc3 weight gradient')
    close(db, dout.double().sum(0), tol=1e-5, what='fc3 bias gradient')


@pytest.mark.parametrize('n,gs,C0,C1,Co,H,W', [(2, 1, 8, 0, 8, 24, 48), (2, 2, 40, 0, 8, 16, 64), (2, 1, 6, 2, 4, 16, 64)])
def test_bf16_math_flag_falls_back_to_fp32_kernels(L, n, gs, C0, C1, Co, H, W):
    """MO_BF_MATH on a shape the bf16 matrix-pipe kernels do not serve (image that does not tile into 16 x 64 pixels, more
    than 32 input channels, a concat of unequal halves): the flag is ignored and the fp32 kernels answer with fp32
    accuracy (1e-4), statistics rows as mo_conv3x3_stats_tiles2 announces them."""
    lib = L.load()
    G = n // gs
    x0 = rand(1, (n, C0, H, W)); sc0, sh0 = rand(2, (G, C0)) * 0.3 + 1, rand(3, (G, C0)) * 0.3
    ins = [act_view(x0, sc0, sh0, gs)]
    x1 = rand(4, (n, C1, H, W)) if C1 else None
    if C1:
        ins.append(x1)
    Ci = C0 + C1
    Wt = rand(5, (Co, Ci, 3, 3)) / np.sqrt(9 * Ci)
    ref = F.conv2d(torch.cat(ins, 1), Wt, None, padding=1)
    out = torch.empty(n, Co, H, W, device='cuda')
    ntile = lib.mo_conv3x3_stats_tiles2(C0, C1, Co, n, H, W, L.BF_MATH)
    stats = torch.full((n, max(ntile, 1), Co, 2), float('nan'), device='cuda')
    L.call('mo_conv3x3_fwd', L.ptr(dev(x0)), C0, C0 * H * W, L.ptr(dev(sc0)), L.ptr(dev(sh0)), 1, L.ptr(dev(x1)) if C1 else None,
           C1, C1 * H * W, None, None, 0, gs, L.ptr(dev(Wt)), Co, n, H, W, L.ptr(out), Co * H * W,
           L.ptr(stats) if ntile else None, L.BF_MATH, None, L.stream())
    close(out, ref, tol=1e-4, what='conv fwd (fp32 fallback under MO_BF_MATH)')
    if ntile:
        close(stats.cpu().double().sum(1)[..., 0], ref.double().sum((2, 3)), tol=1e-5, what='epilogue sum')


@pytest.mark.parametrize('n,gs,Ci,Co,HW,perm', [(4, 2, 4, 1, 256, False), (6, 2, 4, 13, 64 * 64, True), (3, 3, 4, 16, 5000, True),
                                                (2, 1, 3, 3, 64, False), (12, 2, 4, 4, 128 * 128, True)])
@pytest.mark.parametrize('bf', [0, 1])
def test_outconv_loss_fused(L, n, gs, Ci, Co, HW, perm, bf):
    """mo_outc_loss_fwd / _bwd (the tail of training_step, lit.py:32-38 on unet.py:86-92): OutConv + MSE / MAE / MAPE / RMSE,
    the data gradient of the OutConv and its weight / bias gradients from one pass, for an upstream gradient handed in as
    a device scalar; the target addressed through per-image offsets (a permuted batch).  Against torch on the CPU:
    nn.MSELoss, the torchmetrics definitions restated in oracle/metrics_ref.py (parity unpinned for torchmetrics itself)."""
    from oracle import metrics_ref
    lib = L.load()
    G = n // gs
    x = rand(40, (n, Ci, HW, 1))
    x = (x.to(torch.bfloat16).float() if bf else x).requires_grad_(True)
    sc, sh = rand(41, (G, Ci)) * 0.3 + 1, rand(42, (G, Ci)) * 0.3
    a = act_view(x, sc, sh, gs)
    a.retain_grad()
    Wt = rand(43, (Co, Ci, 1, 1)).requires_grad_(True)
    b = rand(44, (Co,)).requires_grad_(True)
    yhat = F.conv2d(a, Wt, b)
    tgt = rand(45, tuple(yhat.shape))
    tgt[0, 0, :5] = 0.0                                   # MAPE's clamp at 1.17e-6
    loss = F.mse_loss(yhat, tgt)
    upstream = 0.37
    (loss * upstream).backward()
    mae, mape, rmse = metrics_ref.metrics(yhat.detach(), tgt)
    # the target as the permuted view of lit.py:31 would hand it: images stored in another order
    if perm:
        order = torch.randperm(n, generator=torch.Generator().manual_seed(7))
        store = torch.empty_like(tgt)
        store[order] = tgt                                # image i lives at slot order[i]
        td = dev(store.reshape(n, Co, HW))
        off = dev((order * (Co * HW)).to(torch.int64))
    else:
        td, off = dev(tgt.reshape(n, Co, HW)), None
    xd, scd, shd = dev(x.detach().to(torch.bfloat16) if bf else x.detach()), dev(sc), dev(sh)
    da = torch.full((n, Ci, HW), float('nan'), device='cuda').to(torch.bfloat16 if bf else torch.float32)
    ws = torch.empty(lib.mo_outc_loss_ws_floats(n, HW, Ci, Co), device='cuda')
    out4 = torch.empty(4, device='cuda')
    yh = torch.empty(n, Co, HW, device='cuda')
    L.call('mo_outc_loss_fwd', L.ptr(xd), Ci * HW, Ci, L.ptr(scd), L.ptr(shd), 1, gs, L.ptr(dev(Wt.detach())),
           L.ptr(dev(b.detach())), Co, L.ptr(td), L.ptr(off), n, HW, L.ptr(yh), L.ptr(da), Ci * HW, L.ptr(ws), L.ptr(out4),
           (L.BF_IN0 | L.BF_OUT) * bf, L.stream())
    close(yh, yhat.reshape(n, Co, HW), what='yhat')
    o = out4.cpu()
    assert abs(o[0] - loss.item()) <= 1e-5 * loss.item() and abs(o[1] - mae.item()) <= 1e-5 * mae.item()
    assert abs(o[2] - mape.item()) <= 1e-4 * mape.item() and abs(o[3] - rmse.item()) <= 1e-5 * rmse.item()
    # unit upstream gradient in da / the slab; the scale arrives as a device scalar
    close(da.float() * upstream, a.grad.reshape(n, Ci, HW), tol=4e-3 if bf else 1e-4, what='da')
    scale = dev(torch.tensor([upstream]))
    dW = torch.empty(Co, Ci, device='cuda'); db = torch.empty(Co, device='cuda')
    L.call('mo_outc_loss_bwd', L.ptr(xd), Ci * HW, Ci, L.ptr(scd), L.ptr(shd), 1, gs, L.ptr(dev(Wt.detach())),
           L.ptr(dev(b.detach())), Co, L.ptr(td), L.ptr(off), n, HW, L.ptr(ws), L.ptr(scale), L.ptr(dW), L.ptr(db),
           L.BF_IN0 * bf, L.stream())
    close(dW, Wt.grad.reshape(Co, Ci), what='dW')
    close(db, b.grad, what='db')
    # without the optional prediction the results are the same bits
    da2 = torch.empty_like(da); out4b = torch.empty(4, device='cuda')
    L.call('mo_outc_loss_fwd', L.ptr(xd), Ci * HW, Ci, L.ptr(scd), L.ptr(shd), 1, gs, L.ptr(dev(Wt.detach())),
           L.ptr(dev(b.detach())), Co, L.ptr(td), L.ptr(off), n, HW, None, L.ptr(da2), Ci * HW, L.ptr(ws), L.ptr(out4b),
           (L.BF_IN0 | L.BF_OUT) * bf, L.stream())
    assert torch.equal(out4, out4b) and torch.equal(da.view(torch.int16 if bf else torch.int32), da2.view(torch.int16 if bf else torch.int32))
