"""GPU parity tests, model level: gwnet (HIP engine through the C-ABI) against the golden vectors of
the reference's own class bodies (tests/golden, tools/make_goldens.py) and against the CPU oracle
on the same seeded inputs.  Tolerance: 1e-4 (fp32 bound of the north star), written per check."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import rand, golden, assert_close, check_grads
from oracle import params as P
from oracle import gwnet_ref

pytestmark = pytest.mark.gpu

GW_CASES = {
    'gwnet_C1': dict(B=4, N=20, T=12, in_dim=2, out_dim=12, K=2, nsup=2, seed=200, knn=(20, 0)),
    'gwnet_C1b': dict(B=3, N=37, T=5, in_dim=5, out_dim=3, K=2, nsup=1, seed=210, knn=(37, 3)),
    'gwnet_C1c': dict(B=2, N=20, T=16, in_dim=4, out_dim=6, K=2, nsup=2, seed=220, knn=(20, 0)),
}


def _supports(cfg):
    A = P.knn_graph(cfg['knn'][0], seed=cfg['knn'][1])
    return [gwnet_ref.asym_adj(A), gwnet_ref.asym_adj(A.T)][:cfg['nsup']]


def _model(cfg, supports, dropout=0.0):
    from multimodal_outage_amd.models.graph_wavenet import gwnet
    m = gwnet('cpu', num_nodes=cfg['N'], dropout=dropout, supports=supports, in_dim=cfg['in_dim'],
              out_dim=cfg['out_dim'], kernel_size=cfg['K'])
    schema = P.gwnet_schema(num_nodes=cfg['N'], supports_len=len(supports) + 1, in_dim=cfg['in_dim'],
                            out_dim=cfg['out_dim'], kernel_size=cfg['K'])
    P.load_into(m, P.seeded_values(schema, cfg['seed']))
    return m.cuda()


@pytest.mark.parametrize('name', list(GW_CASES))
def test_gwnet_generic_vs_golden(name):
    cfg = GW_CASES[name]
    G = golden(name)
    m = _model(cfg, _supports(cfg))
    m.train()
    x = rand(cfg['seed'] + 1, (cfg['B'], cfg['in_dim'], cfg['N'], cfg['T'])).cuda().requires_grad_(True)
    y = m(x)
    assert_close(y, G['y'], 1e-4, 1e-4, 'y')
    tgt = rand(cfg['seed'] + 2, tuple(y.shape)).cuda()
    loss = F.mse_loss(y, tgt)
    assert abs(loss.item() - float(G['loss'])) < 1e-4 * float(G['loss'])
    loss.backward()
    assert_close(x.grad, G['dx'], 1e-6, 1e-3, 'dx')
    check_grads({k: v.grad for k, v in m.named_parameters()}, G, atol=2e-6, rtol=1e-3)
    sd = m.state_dict()
    for k in G.files:
        if k.startswith('buf/'):
            assert_close(sd[k[4:]].float(), G[k], 1e-5, 1e-4, k)
    m.eval()
    with torch.no_grad():
        ye = m(x.detach())
    assert_close(ye, G['y_eval'], 1e-4, 1e-4, 'y_eval')


def test_gwnet_reference_views_vs_golden():
    """R config: the (67,7,320) input through the raw views of graph_wavenet.py:189/:255."""
    from multimodal_outage_amd.models.graph_wavenet import gwnet
    G = golden('gwnet_R')
    m = gwnet('cpu', in_dim=320, out_dim=256, horizon=7, dropout=0.0)
    schema = P.gwnet_schema(num_nodes=67, supports_len=2, in_dim=320, out_dim=256, kernel_size=1)
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == [(k, tuple(v)) for k, v in schema.items()]
    P.load_into(m, P.seeded_values(schema, 100))
    m = m.cuda().train()
    x = rand(101, (67, 7, 320)).cuda().requires_grad_(True)
    y = m(x)
    assert tuple(y.shape) == (67, 7, 256)
    assert_close(y, G['y'], 1e-4, 1e-4, 'y')
    loss = F.mse_loss(y, rand(102, (67, 7, 256)).cuda())
    assert abs(loss.item() - float(G['loss'])) < 1e-4
    loss.backward()
    # input gradient: fp32 summation-order noise of the reference itself is ~3e-3 of the tensor scale (fp64 yardstick,
    # tests/helpers.check_grads_vs_f64), so the absolute term is stated relative to max|dx| (6.3e-5 here)
    assert_close(x.grad, G['dx'], 5e-3 * float(np.abs(G['dx']).max()), 1e-3, 'dx')
    check_grads({k: v.grad for k, v in m.named_parameters()}, G, atol=2e-6, rtol=1e-3)
    sd = m.state_dict()
    for k in G.files:
        if k.startswith('buf/'):
            assert_close(sd[k[4:]].float(), G[k], 1e-5, 1e-4, k)


def test_gwnet_vs_oracle_midsize():
    """N=300 nodes, B=2, T=12, C=32, K=2, 2 static + adaptive supports: HIP vs the CPU oracle on the
    same seeded inputs (the oracle finishes in seconds at this size)."""
    cfg = dict(B=2, N=300, T=12, in_dim=32, out_dim=12, K=2, nsup=2, seed=900, knn=(300, 1))
    sup = _supports(cfg)
    m = _model(cfg, sup).train()
    schema = P.gwnet_schema(num_nodes=300, supports_len=3, in_dim=32, out_dim=12, kernel_size=2)
    p = P.as_param_dict(P.seeded_values(schema, 900))
    x = rand(901, (2, 32, 300, 12))
    xr = x.clone().requires_grad_(True)
    yr = gwnet_ref.gwnet_forward(p, xr, supports=[torch.from_numpy(s) for s in sup], kernel_size=2)
    tgt = rand(902, tuple(yr.shape))
    F.mse_loss(yr, tgt).backward()
    xg = x.cuda().requires_grad_(True)
    y = m(xg)
    assert_close(y, yr.detach(), 1e-4, 1e-4, 'y')
    F.mse_loss(y, tgt.cuda()).backward()
    assert_close(xg.grad, xr.grad, 1e-7, 2e-3, 'dx')
    for k, v in m.named_parameters():
        if p[k].grad is None:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0
            continue
        scale = float(p[k].grad.abs().max())
        err = float((v.grad.cpu() - p[k].grad).abs().max())
        assert err <= 1e-3 * scale + 1e-7, (k, err, scale)


def test_gwnet_flat_trainer_steps_follow_oracle_adam():
    """Four optimizer steps of the product path (renumbered node space, gradients written into the flat buffer or
    accumulated there by autograd for the node embeddings, fused Adam on the flat buffers) against the CPU oracle driven
    by torch.optim.Adam(lr=1e-3) (lit.py:59-61) on the same seeded parameters and batch: the loss trajectory must agree
    (each loss depends on every earlier update), and so must the parameters that received real gradients."""
    from multimodal_outage_amd.trainer import FlatTrainer
    cfg = dict(B=2, N=60, T=12, in_dim=8, out_dim=12, K=2, nsup=2, seed=930, knn=(60, 4))
    sup = _supports(cfg)
    m = _model(cfg, sup).train()
    schema = P.gwnet_schema(num_nodes=60, supports_len=3, in_dim=8, out_dim=12, kernel_size=2)
    p = P.as_param_dict(P.seeded_values(schema, 930))
    tr = FlatTrainer(m, lr=1e-3)
    m._mo_grad_out = tr.grad_out()
    opt = torch.optim.Adam([v for v in p.values()], lr=1e-3)
    x = rand(931, (2, 8, 60, 12))
    tgt = None
    sups = [torch.from_numpy(s) for s in sup]
    for step in range(4):
        opt.zero_grad()
        yr = gwnet_ref.gwnet_forward(p, x, supports=sups, kernel_size=2)
        if tgt is None:
            tgt = rand(932, tuple(yr.shape))
        lr_ = F.mse_loss(yr, tgt)
        lr_.backward()
        opt.step()
        tr.zero_grad()
        loss = F.mse_loss(m(x.cuda()), tgt.cuda())
        loss.backward()
        tr.allreduce()
        tr.step()
        lh, lo = float(loss.detach()), float(lr_.detach())
        assert abs(lh - lo) <= 2e-4 * abs(lo), (step, lh, lo)
    got = dict(m.named_parameters())
    for k in ('start_conv.weight', 'end_conv_2.bias', 'gconv.2.mlp.mlp.weight', 'filter_convs.4.weight', 'bn.1.weight',
              'skip_convs.6.weight', 'nodevec1', 'nodevec2'):
        a, b = got[k].detach().cpu(), p[k].detach()
        # an Adam step moves every coordinate by ~lr whatever the gradient's size, so coordinates whose gradient is at
        # rounding level may legitimately differ by a few lr; the bulk must agree far below one step
        d = (a - b).abs()
        assert float(d.median()) <= 2e-5 and float((d > 5e-4).float().mean()) <= 0.02, (k, float(d.median()), float(d.max()))


def test_gwnet_directed_supports_vs_oracle():
    """Two genuinely different, asymmetric supports (a directed graph and its reverse with unequal weights): the node
    renumbering, the forward (A^T) and backward (A) CSR forms and their block-union variants must all agree with the
    CPU oracle in the original node order -- outputs, input gradient and every parameter gradient (both modes)."""
    N = 96
    rng = np.random.RandomState(5)
    A = P.knn_graph(N, seed=6).astype(np.float64)
    A = A * (rng.rand(N, N) < 0.6) * (0.5 + rng.rand(N, N))            # drop 40 % of the directions, unequal weights
    sup = [gwnet_ref.asym_adj(A).astype(np.float32), gwnet_ref.asym_adj(A.T).astype(np.float32)]
    assert not np.allclose(sup[0], sup[1]) and not np.allclose(sup[0], sup[0].T)
    cfg = dict(B=2, N=N, T=12, in_dim=8, out_dim=12, K=2, nsup=2, seed=940, knn=(N, 6))
    schema = P.gwnet_schema(num_nodes=N, supports_len=3, in_dim=8, out_dim=12, kernel_size=2)
    p = P.as_param_dict(P.seeded_values(schema, 940))
    x = rand(941, (2, 8, N, 12))
    xr = x.clone().requires_grad_(True)
    yr = gwnet_ref.gwnet_forward(p, xr, supports=[torch.from_numpy(s_) for s_ in sup], kernel_size=2)
    tgt = rand(942, tuple(yr.shape))
    F.mse_loss(yr, tgt).backward()
    for mode, tol_y, tol_g in (('f32', 1e-4, 1e-3), ('bf16', 2e-2, 1e-1)):
        m = _model(cfg, sup).train()
        m.dense_dtype = mode
        xg = x.cuda().requires_grad_(True)
        y = m(xg)
        assert float((y.detach().cpu() - yr.detach()).abs().max()) <= tol_y * float(yr.abs().max()) + 1e-6, mode
        F.mse_loss(y, tgt.cuda()).backward()
        assert float((xg.grad.cpu() - xr.grad).abs().max()) <= tol_g * float(xr.grad.abs().max()) * 5 + 1e-7, mode
        for k, v in m.named_parameters():
            if p[k].grad is None:
                continue
            scale = float(p[k].grad.abs().max())
            err = float((v.grad.cpu() - p[k].grad).abs().max())
            # the node embeddings' gradient (a sum over all layers of products of bf16-rounded tensors pushed through
            # the softmax) is the noisiest tensor of the throughput mode: 2e-1 of its scale here, 1e-1 for the rest
            lim = 2 * tol_g if (mode == 'bf16' and k.startswith('nodevec')) else tol_g
            assert err <= lim * scale + 1e-7, (mode, k, err, scale)


def test_gwnet_dropout_training_statistics():
    """dropout=0.3 (graph_wavenet.py:97): own counter-based mask; bitwise RNG parity with CPU torch is
    unattainable, so check determinism of backward w.r.t. the forward mask via a finite-difference
    style identity: loss is differentiable and grads are finite; eval() ignores dropout."""
    cfg = GW_CASES['gwnet_C1']
    m = _model(cfg, _supports(cfg), dropout=0.3).train()
    x = rand(5, (4, 2, 20, 12)).cuda()
    torch.manual_seed(1)
    y1 = m(x)
    torch.manual_seed(1)
    y2 = m(x)
    assert torch.equal(y1, y2)                       # same host seed -> same mask
    y3 = m(x)
    assert not torch.equal(y1, y3)
    y1.square().mean().backward()
    for k, v in m.named_parameters():
        if v.grad is not None:
            assert torch.isfinite(v.grad).all(), k
    m.eval()
    with torch.no_grad():
        assert torch.equal(m(x), m(x))


def test_gwnet_config2_shape_properties():
    """BASELINE config 2 size (N=3000, C=32, T=12, K=2, S=3) at B=2: size-independent properties --
    output shape, finiteness, linearity of the head in the skip path is not available, so check
    (a) batch independence of eval-mode outputs and (b) gradient of sum(y) w.r.t. the last bias."""
    N = 3000
    A = P.knn_graph(N)
    sup = [gwnet_ref.asym_adj(A), gwnet_ref.asym_adj(A.T)]
    from multimodal_outage_amd.models.graph_wavenet import gwnet
    torch.manual_seed(42)
    m = gwnet('cpu', num_nodes=N, dropout=0.0, supports=sup, in_dim=32, out_dim=12, kernel_size=2).cuda()
    x = rand(7, (2, 32, N, 12)).cuda()
    m.train()
    y = m(x)
    assert tuple(y.shape) == (2, 12, N, 1) and torch.isfinite(y).all()
    y.sum().backward()
    # d sum(y) / d end_conv_2.bias[c] = number of output positions per channel
    assert_close(m.end_conv_2.bias.grad, np.full(12, 2.0 * N, dtype=np.float32), 1e-2, 1e-5)
    m.eval()
    with torch.no_grad():
        ya = m(x)
        yb = m(x[:1])
    assert_close(ya[:1], yb, 1e-5, 1e-4, 'eval-mode batch independence')


def test_gwnet_bf16_dense_mode_close_to_fp32():
    """Throughput mode (BASELINE config 2, bf16): bf16 operands (fp32 accumulate) for the dense adaptive-adjacency
    products, and the diffusion intermediates x1/x2 and their gradients STORED as bf16 tensors; every other
    contraction stays fp32.  Stated tolerance vs the fp32 mode on the same inputs: 2e-2 of the output scale, loss
    within 1e-2 relative; gradients within 1e-1 of each tensor's scale (the node embeddings' gradient, a sum over all
    layers of products of bf16-rounded tensors pushed through the softmax, is the noisiest at ~6e-2)."""
    cfg = dict(B=2, N=304, T=12, in_dim=32, out_dim=12, K=2, nsup=2, seed=910, knn=(304, 1))
    sup = _supports(cfg)
    outs = {}
    for mode in ('f32', 'bf16'):
        m = _model(cfg, sup).train()
        m.dense_dtype = mode
        x = rand(911, (2, 32, 304, 12)).cuda().requires_grad_(True)
        y = m(x)
        loss = F.mse_loss(y, rand(912, tuple(y.shape)).cuda())
        loss.backward()
        outs[mode] = (y.detach(), loss.item(), {k: v.grad.clone() for k, v in m.named_parameters() if v.grad is not None})
    y32, l32, g32 = outs['f32']
    y16, l16, g16 = outs['bf16']
    assert float((y16 - y32).abs().max()) <= 2e-2 * float(y32.abs().max())
    assert abs(l16 - l32) <= 1e-2 * l32
    for k in g32:
        s = float(g32[k].abs().max())
        assert float((g16[k] - g32[k]).abs().max()) <= 1e-1 * s + 1e-7, k


def test_gwnet_throughput_mode_full_size_close_to_fp32():
    """BASELINE config 2 size (N=3000, T=12, C=32), dropout on: the throughput mode (bf16 storage + bf16 MFMA paths of
    every row-streaming kernel, the bf16 head GEMMs, ragged 128-row runs at P = 3000*3*T') against the fp32 mode with
    the same dropout masks (the mask is a pure function of seed and element index).  Same stated tolerances as the
    small case; gradients of a few representative tensors."""
    cfg = dict(B=3, N=3000, T=12, in_dim=32, out_dim=12, K=2, nsup=2, seed=920, knn=(3000, 2))
    sup = _supports(cfg)
    outs = {}
    for mode in ('f32', 'bf16'):
        m = _model(cfg, sup, dropout=0.3).train()
        m.dense_dtype = mode
        torch.manual_seed(5)
        x = rand(921, (3, 32, 3000, 12)).cuda()
        y = m(x)
        loss = F.mse_loss(y, rand(922, tuple(y.shape)).cuda())
        loss.backward()
        keys = ('end_conv_1.weight', 'skip_convs.0.weight', 'gconv.3.mlp.mlp.weight', 'filter_convs.5.weight',
                'filter_convs.5.bias', 'start_conv.weight', 'bn.2.weight')
        g = dict(m.named_parameters())
        outs[mode] = (y.detach(), loss.item(), {k: g[k].grad.clone() for k in keys})
    y32, l32, g32 = outs['f32']
    y16, l16, g16 = outs['bf16']
    assert float((y16 - y32).abs().max()) <= 2e-2 * float(y32.abs().max())
    assert abs(l16 - l32) <= 1e-2 * l32
    for k in g32:
        s = float(g32[k].abs().max())
        assert float((g16[k] - g32[k]).abs().max()) <= 1e-1 * s + 1e-7, k


VARIANTS = {
    'gwnet_V_nogcn': dict(B=3, N=20, T=12, in_dim=3, out_dim=4, K=2, seed=230, gcn_bool=False, addaptadj=True),
    'gwnet_V_static': dict(B=3, N=20, T=12, in_dim=3, out_dim=4, K=2, seed=240, gcn_bool=True, addaptadj=False),
    'gwnet_V_k1': dict(B=2, N=20, T=7, in_dim=6, out_dim=5, K=1, seed=250, gcn_bool=True, addaptadj=True),
}


@pytest.mark.parametrize('name', list(VARIANTS))
def test_gwnet_constructor_variants_vs_golden(name):
    """Constructor variants of graph_wavenet.py:101: gcn_bool=False (residual_convs), addaptadj=False, K=1."""
    from multimodal_outage_amd.models.graph_wavenet import gwnet
    cfg = VARIANTS[name]
    G = golden(name)
    A = P.knn_graph(20)
    sup = [gwnet_ref.asym_adj(A), gwnet_ref.asym_adj(A.T)]
    adaptive = cfg['gcn_bool'] and cfg['addaptadj']
    m = gwnet('cpu', num_nodes=20, dropout=0.0, supports=sup, in_dim=cfg['in_dim'], out_dim=cfg['out_dim'],
              kernel_size=cfg['K'], gcn_bool=cfg['gcn_bool'], addaptadj=cfg['addaptadj'])
    schema = P.gwnet_schema(num_nodes=20, supports_len=2 + (1 if adaptive else 0), in_dim=cfg['in_dim'],
                            out_dim=cfg['out_dim'], kernel_size=cfg['K'], gcn_bool=cfg['gcn_bool'],
                            addaptadj=cfg['addaptadj'])
    P.load_into(m, P.seeded_values(schema, cfg['seed']))
    m = m.cuda().train()
    x = rand(cfg['seed'] + 1, (cfg['B'], cfg['in_dim'], 20, cfg['T'])).cuda().requires_grad_(True)
    y = m(x)
    assert_close(y, G['y'], 1e-4, 1e-4, 'y')
    loss = F.mse_loss(y, rand(cfg['seed'] + 2, tuple(y.shape)).cuda())
    assert abs(loss.item() - float(G['loss'])) < 1e-4 * float(G['loss'])
    loss.backward()
    assert_close(x.grad, G['dx'], 1e-6, 1e-3, 'dx')
    check_grads({k: v.grad for k, v in m.named_parameters()}, G, atol=2e-6, rtol=1e-3)
    sd = m.state_dict()
    for k in G.files:
        if k.startswith('buf/'):
            assert_close(sd[k[4:]].float(), G[k], 1e-5, 1e-4, k)


def test_gwnet_supports_none_adaptive_only_vs_oracle():
    """supports=None with addaptadj=True (graph_wavenet.py:130-133): the adaptive adjacency is the only support."""
    from multimodal_outage_amd.models.graph_wavenet import gwnet
    N = 24
    m = gwnet('cpu', num_nodes=N, dropout=0.0, supports=None, in_dim=4, out_dim=3, kernel_size=2)
    assert m.supports_len == 1 and m.gconv[0].mlp.mlp.weight.shape[1] == 3 * 32
    schema = P.gwnet_schema(num_nodes=N, supports_len=1, in_dim=4, out_dim=3, kernel_size=2)
    vals = P.seeded_values(schema, 260)
    P.load_into(m, vals)
    m = m.cuda().train()
    p = P.as_param_dict(vals)
    x = rand(261, (2, 4, N, 12))
    xr = x.clone().requires_grad_(True)
    yr = gwnet_ref.gwnet_forward(p, xr, supports=[], kernel_size=2)
    tgt = rand(262, tuple(yr.shape))
    F.mse_loss(yr, tgt).backward()
    xg = x.cuda().requires_grad_(True)
    y = m(xg)
    assert_close(y, yr.detach(), 1e-4, 1e-4, 'y')
    F.mse_loss(y, tgt.cuda()).backward()
    assert_close(xg.grad, xr.grad, 1e-6, 1e-3, 'dx')
    for k, v in m.named_parameters():
        if p[k].grad is not None:
            s = float(p[k].grad.abs().max())
            assert float((v.grad.cpu() - p[k].grad).abs().max()) <= 1e-3 * s + 1e-7, k
