"""Pin the CPU oracle (oracle/) against the golden vectors produced by the reference's own class
bodies (tools/make_goldens.py).  CPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import params as P
from oracle import gwnet_ref, unet_ref, metrics_ref
from helpers import rand, golden, assert_close, check_grads

GW_CASES = {
    'gwnet_C1': dict(B=4, N=20, T=12, in_dim=2, out_dim=12, K=2, nsup=2, seed=200, knn=(20, 0)),
    'gwnet_C1b': dict(B=3, N=37, T=5, in_dim=5, out_dim=3, K=2, nsup=1, seed=210, knn=(37, 3)),
    'gwnet_C1c': dict(B=2, N=20, T=16, in_dim=4, out_dim=6, K=2, nsup=2, seed=220, knn=(20, 0)),
}


def _supports(cfg):
    A = P.knn_graph(cfg['knn'][0], seed=cfg['knn'][1])
    s = [gwnet_ref.asym_adj(A), gwnet_ref.asym_adj(A.T)][:cfg['nsup']]
    return [torch.from_numpy(x) for x in s]


@pytest.mark.parametrize('name', list(GW_CASES))
def test_gwnet_generic(name):
    cfg = GW_CASES[name]
    G = golden(name)
    schema = P.gwnet_schema(num_nodes=cfg['N'], supports_len=cfg['nsup'] + 1, in_dim=cfg['in_dim'],
                            out_dim=cfg['out_dim'], kernel_size=cfg['K'])
    p = P.as_param_dict(P.seeded_values(schema, cfg['seed']))
    x = rand(cfg['seed'] + 1, (cfg['B'], cfg['in_dim'], cfg['N'], cfg['T'])).requires_grad_(True)
    y = gwnet_ref.gwnet_forward(p, x, supports=_supports(cfg), kernel_size=cfg['K'])
    assert_close(y, G['y'], 1e-5, 1e-5, 'y')
    tgt = rand(cfg['seed'] + 2, tuple(y.shape))
    loss = F.mse_loss(y, tgt)
    assert abs(loss.item() - float(G['loss'])) < 1e-6
    loss.backward()
    assert_close(x.grad, G['dx'], 1e-6, 1e-4, 'dx')
    grads = {k: v.grad for k, v in p.items() if v.requires_grad}
    check_grads(grads, G, atol=1e-6, rtol=1e-4)
    for k in G.files:
        if k.startswith('buf/'):
            assert_close(p[k[4:]], G[k], 1e-6, 1e-5, k)
    assert int(G['receptive_field']) == gwnet_ref.receptive_field(cfg['K'], 4, 2)
    with torch.no_grad():
        assert_close(gwnet_ref.adaptive_adj(p['nodevec1'], p['nodevec2']), G['adp'], 1e-7, 1e-5)
        ye = gwnet_ref.gwnet_forward(p, x.detach(), supports=_supports(cfg),
                                     kernel_size=cfg['K'], training=False)
    assert_close(ye, G['y_eval'], 1e-5, 1e-5, 'y_eval')


def test_gwnet_reference_views():
    """R config: (67,7,320) through the raw view of graph_wavenet.py:189/:255, K=1, [I_67]."""
    G = golden('gwnet_R')
    schema = P.gwnet_schema(num_nodes=67, supports_len=2, in_dim=320, out_dim=256, kernel_size=1)
    p = P.as_param_dict(P.seeded_values(schema, 100))
    x = rand(101, (67, 7, 320)).requires_grad_(True)
    y = gwnet_ref.gwnet_forward_ref_views(p, x, horizon=7, supports=[torch.eye(67)])
    assert_close(y, G['y'], 1e-5, 1e-5, 'y')
    loss = F.mse_loss(y, rand(102, (67, 7, 256)))
    assert abs(loss.item() - float(G['loss'])) < 1e-6
    loss.backward()
    assert_close(x.grad, G['dx'], 1e-7, 1e-4, 'dx')
    check_grads({k: v.grad for k, v in p.items() if v.requires_grad}, G, atol=1e-6, rtol=1e-4)
    none = sorted(str(s) for s in G['none_grads'])
    # residual_convs.*, gconv.7, bn.7 never receive a gradient (13,664 params)
    assert len(none) == 20 and sum(int(np.prod(schema[k])) for k in none) == 13664


def test_adjacency_and_csr():
    import scipy.sparse as sp
    G = golden('adjacency')
    A = G['adj']
    assert A.shape == (67, 67) and int(A.sum()) == 312 and (A == A.T).all()
    c = sp.csr_matrix(A)
    assert (c.indptr == G['rowptr']).all() and (c.indices == G['colidx']).all()
    assert_close(gwnet_ref.asym_adj(A), G['asym'], 0, 0)
    assert_close(gwnet_ref.asym_adj(A.T), G['asym_t'], 0, 0)
    assert (G['load_adj0'] == np.eye(67, dtype=np.float32)).all()
    assert (P.knn_graph(20) == G['knn20']).all()
    assert_close(gwnet_ref.asym_adj(G['knn20']), G['knn20_asym'], 0, 0)


def test_date2vec():
    G = golden('date2vec')
    keys = [str(k) for k in G['keys']]
    shapes = {'fc1': (32, 6), 'fc2': (32, 6), 'fc3': (32, 64), 'fc4': (6, 32), 'fc5': (6, 6)}
    schema = {}
    for k in keys:
        mod, kind = k.split('.')
        schema[k] = shapes[mod] if kind == 'weight' else (shapes[mod][0],)
    v = P.seeded_values(schema, int(G['seed']))
    y = unet_ref.date2vec_encode(torch.from_numpy(G['x']), v['fc1.weight'], v['fc1.bias'],
                                 v['fc2.weight'], v['fc2.bias'])
    assert_close(y, G['y'], 1e-5, 1e-5)


def _block_params(G, nm):
    keys = [str(k) for k in G[nm + '/keys']]
    return keys


def test_unet_blocks():
    G = golden('unet_blocks')
    seed = 300

    def vals(nm, shapes):
        keys = [str(k) for k in G[nm + '/keys']]
        return P.as_param_dict(P.seeded_values({k: shapes[k] for k in keys}, seed))

    def dc_shapes(pre, ci, co):
        d = {pre + 'double_conv.0.weight': (co, ci, 3, 3), pre + 'double_conv.3.weight': (co, co, 3, 3)}
        for j in (1, 4):
            for s in ('weight', 'bias', 'running_mean', 'running_var'):
                d[pre + f'double_conv.{j}.{s}'] = (co,)
            d[pre + f'double_conv.{j}.num_batches_tracked'] = ()
        return d

    def run(nm, p, fn, ins):
        ins = [t.requires_grad_(True) for t in ins]
        y = fn(p, *ins)
        assert_close(y, G[nm + '/y'], 1e-5, 1e-5, nm)
        loss = F.mse_loss(y, rand(seed + 20, tuple(y.shape)))
        assert abs(loss.item() - float(G[nm + '/loss'])) < 1e-6
        loss.backward()
        for i, t in enumerate(ins):
            assert_close(t.grad, G[f'{nm}/dx{i}'], 1e-7, 1e-4, f'{nm} dx{i}')
        for k, v in p.items():
            if v.requires_grad:
                assert_close(v.grad, G[f'{nm}/grad/{k}'], 1e-7, 1e-4, f'{nm} {k}')
            elif f'{nm}/buf/{k}' in G.files:
                assert_close(v, G[f'{nm}/buf/{k}'], 1e-6, 1e-5, f'{nm} {k}')

    p = vals('double_conv', dc_shapes('', 3, 8))
    run('double_conv', p, lambda p, x: unet_ref.double_conv(_Pre(p, 'm.'), 'm', x, True),
        [rand(seed + 10, (2, 3, 12, 12))])
    p = vals('down', dc_shapes('maxpool_conv.1.', 4, 8))
    run('down', p, lambda p, x: unet_ref.down(_Pre(p, 'm.'), 'm', x, True),
        [rand(seed + 10, (3, 4, 16, 16))])
    sh = dc_shapes('conv.', 16, 8)
    sh.update({'up.weight': (16, 8, 2, 2), 'up.bias': (8,)})
    p = vals('up', sh)
    run('up', p, lambda p, a, b: unet_ref.up(_Pre(p, 'm.'), 'm', a, b, True),
        [rand(seed + 10, (2, 16, 6, 8)), rand(seed + 11, (2, 8, 12, 16))])
    p = vals('outc', {'conv.weight': (2, 4, 1, 1), 'conv.bias': (2,)})
    run('outc', p, lambda p, x: F.conv2d(x, p['conv.weight'], p['conv.bias']),
        [rand(seed + 10, (2, 4, 8, 8))])
    # composite blocks (unet.py:95-199): Contraction -> Encoder -> Decoder -> Expansion, 3 counties x 2 days of 64x64
    NC, H, S = 3, 2, 64
    p = {}
    for nm, sd in zip(('contraction', 'encoder', 'decoder', 'expansion'), G['composite/seeds']):
        keys = [str(k) for k in G[nm + '/keys']]
        shp = {k: tuple(G[f'{nm}/grad/{k}'].shape) if f'{nm}/grad/{k}' in G.files else
               (tuple(G[f'{nm}/buf/{k}'].shape)) for k in keys}
        for k, v in P.as_param_dict(P.seeded_values(shp, int(sd))).items():
            p[f'{nm}.{k}'] = v
    x = rand(seed + 30, (NC, H, 2, S, S)).requires_grad_(True)
    feat, fms = unet_ref.contraction(p, x, H, True)
    assert_close(feat, G['composite/feat'], 1e-5, 1e-5, 'contraction')
    for k in range(4):
        assert_close(fms[k], G[f'composite/fm{k}'], 1e-5, 1e-5, f'fm{k}')
    z = unet_ref.fc_block(p, 'encoder', feat, 0.0, True)
    assert_close(z, G['composite/z'], 1e-5, 1e-5, 'encoder')
    e = unet_ref.fc_block(p, 'decoder', z, 0.0, True).view(NC, H, 64, 4, 4)
    assert_close(e, G['composite/e'], 1e-5, 1e-5, 'decoder')
    y = unet_ref.expansion(p, e, fms, True)
    assert_close(y, G['composite/y'], 1e-5, 1e-5, 'expansion')
    loss = F.mse_loss(y, rand(seed + 31, tuple(y.shape)))
    assert abs(loss.item() - float(G['composite/loss'])) < 1e-6
    loss.backward()
    assert_close(x.grad, G['composite/dx'], 1e-7, 1e-3, 'composite dx')
    for k, v in p.items():
        nm, kk = k.split('.', 1)
        if v.requires_grad:
            assert_close(v.grad, G[f'{nm}/grad/{kk}'], 1e-6, 2e-3, k)
        elif f'{nm}/buf/{kk}' in G.files:
            assert_close(v, G[f'{nm}/buf/{kk}'], 1e-6, 1e-5, k)


class _Pre:
    """Expose dict d under an added key prefix."""

    def __init__(self, d, pre):
        self.d, self.pre = d, pre

    def __getitem__(self, k):
        assert k.startswith(self.pre)
        return self.d[k[len(self.pre):]]


def test_modified_unet_full():
    """Full Modified_UNET fwd + MSE + bwd, B=2, H=2 (268 tiles of 1x128x128)."""
    torch.set_num_threads(8)
    G = golden('modified_unet_B2H2')
    schema = P.unet_schema()
    assert sum(int(np.prod(s)) for k, s in schema.items()
               if 'running' not in k and 'num_batches' not in k) == 9450497
    p = P.as_param_dict(P.seeded_values(schema, 400))
    x = rand(401, (2, 67, 2, 1, 128, 128))
    tdim = rand(403, (2, 67, 2, 64))
    y = unet_ref.modified_unet_forward(p, x, tdim, horizon=2, supports=[torch.eye(67)])
    assert tuple(y.shape) == tuple(G['y_shape'])
    yn = y.detach().numpy()
    assert_close(yn.reshape(-1)[::997], G['y_sample'], 2e-5, 1e-4, 'y_sample')
    assert_close(yn[0, 0, 0, 0], G['y_first'], 2e-5, 1e-4)
    loss = F.mse_loss(y, rand(402, tuple(y.shape)))
    assert abs(loss.item() - float(G['loss'])) < 1e-5
    loss.backward()
    check_grads({k: v.grad for k, v in p.items() if v.requires_grad}, G, atol=1e-6, rtol=2e-3)
    for k in G.files:
        if k.startswith('buf/'):
            assert_close(p[k[4:]], G[k], 1e-5, 1e-4, k)
    # F7: BN running stats get 67*B sequential updates per step
    assert int(p['contraction.inc.double_conv.1.num_batches_tracked']) == 67 * 2


def test_modified_unet_config3_forward():
    """BASELINE config 3 shape: Modified_UNET on (1,67,2,13,256,256) tiles, FC bottleneck 16384->4096->256 (the
    reference's Encoder/Decoder with image_dimension=256, unet.py:128-136,151-160): oracle forward + MSE against the
    golden of the reference's own class bodies (the backward of this size is covered on the GPU box, where the HIP
    path is checked against the same golden incl. gradients)."""
    torch.set_num_threads(8)
    G = golden('modified_unet_C3')
    schema = P.unet_schema(input_channels=13, output_channels=13, image_dimension=256)
    p = P.as_param_dict(P.seeded_values(schema, 410), requires_grad=False)
    x = rand(411, (1, 67, 2, 13, 256, 256))
    tdim = rand(413, (1, 67, 2, 64))
    with torch.no_grad():
        y = unet_ref.modified_unet_forward(p, x, tdim, horizon=2, supports=[torch.eye(67)])
    assert tuple(y.shape) == tuple(G['y_shape'])
    yn = y.numpy()
    assert_close(yn.reshape(-1)[::997], G['y_sample'], 2e-5, 1e-4, 'y_sample')
    assert_close(yn[-1, -1, -1, -1], G['y_last'], 2e-5, 1e-4)
    loss = F.mse_loss(y, rand(412, tuple(y.shape)))
    assert abs(loss.item() - float(G['loss'])) < 1e-5


def test_metrics_restatement():
    y = rand(1, (5, 7))
    yh = rand(2, (5, 7))
    mae, mape, rmse = metrics_ref.metrics(yh, y)
    assert abs(rmse.item() ** 2 - metrics_ref.mse(yh, y).item()) < 1e-6
    assert abs(mae.item() - (yh - y).abs().mean().item()) < 1e-7
    assert mape.item() > 0


VARIANTS = {
    'gwnet_V_nogcn': dict(B=3, N=20, T=12, in_dim=3, out_dim=4, K=2, seed=230, gcn_bool=False, addaptadj=True),
    'gwnet_V_static': dict(B=3, N=20, T=12, in_dim=3, out_dim=4, K=2, seed=240, gcn_bool=True, addaptadj=False),
    'gwnet_V_k1': dict(B=2, N=20, T=7, in_dim=6, out_dim=5, K=1, seed=250, gcn_bool=True, addaptadj=True),
    'gwnet_V_s4': dict(B=2, N=20, T=12, in_dim=3, out_dim=4, K=2, seed=260, gcn_bool=True, addaptadj=True, nstatic=3),
    'gwnet_V_s5': dict(B=2, N=20, T=6, in_dim=5, out_dim=2, K=2, seed=270, gcn_bool=True, addaptadj=True, nstatic=4),
}


@pytest.mark.parametrize('name', list(VARIANTS))
def test_gwnet_constructor_variants(name):
    """gcn_bool=False (residual_convs path, graph_wavenet.py:245), addaptadj=False (:242-243), kernel_size=1."""
    cfg = VARIANTS[name]
    G = golden(name)
    A = P.knn_graph(20)
    B2 = P.knn_graph(20, seed=11)
    sup = [torch.from_numpy(gwnet_ref.asym_adj(m_)) for m_ in (A, A.T, B2, B2.T)][:cfg.get('nstatic', 2)]
    adaptive = cfg['gcn_bool'] and cfg['addaptadj']
    schema = P.gwnet_schema(num_nodes=20, supports_len=len(sup) + (1 if adaptive else 0), in_dim=cfg['in_dim'],
                            out_dim=cfg['out_dim'], kernel_size=cfg['K'], gcn_bool=cfg['gcn_bool'],
                            addaptadj=cfg['addaptadj'])
    p = P.as_param_dict(P.seeded_values(schema, cfg['seed']))
    x = rand(cfg['seed'] + 1, (cfg['B'], cfg['in_dim'], 20, cfg['T'])).requires_grad_(True)
    y = gwnet_ref.gwnet_forward(p, x, supports=sup, kernel_size=cfg['K'], gcn_bool=cfg['gcn_bool'],
                                addaptadj=cfg['addaptadj'])
    assert_close(y, G['y'], 1e-5, 1e-5, 'y')
    loss = F.mse_loss(y, rand(cfg['seed'] + 2, tuple(y.shape)))
    assert abs(loss.item() - float(G['loss'])) < 1e-6
    loss.backward()
    assert_close(x.grad, G['dx'], 1e-6, 1e-4, 'dx')
    check_grads({k: v.grad for k, v in p.items() if v.requires_grad}, G, atol=1e-6, rtol=1e-4)
