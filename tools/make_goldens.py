"""Generate tests/golden/*.npz by running the reference's own class bodies on CPU.

Run in the build container only:  python tools/make_goldens.py
Reads /root/reference at run time (tools/ref_loader.py); commits only numeric outputs.
Weights/inputs are regenerated in the tests from numpy seeds (oracle/params.py), so fixtures hold
outputs, losses, gradients (full for small tensors, else norm + strided sample) and buffers.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_loader as R                     # noqa: E402
from oracle import params as P             # noqa: E402
from oracle.gwnet_ref import asym_adj      # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
torch.set_num_threads(8)


def rand(seed, shape):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32))


def grad_summary(module, max_full=4096):
    out = {}
    none = []
    for k, prm in module.named_parameters():
        if prm.grad is None:
            none.append(k)
            continue
        g = prm.grad.detach().numpy()
        out['gnorm/' + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
        if g.size <= max_full:
            out['grad/' + k] = g
        else:
            out['gsample/' + k] = g.reshape(-1)[::max(1, g.size // 2048)][:2048].copy()
    out['none_grads'] = np.array(none)
    return out


def buffers(module):
    return {'buf/' + k: v.detach().numpy().copy() for k, v in module.state_dict().items()
            if 'running_' in k or 'num_batches' in k}


def schema_of(module):
    return [(k, tuple(v.shape)) for k, v in module.state_dict().items()]


def gwnet_case(name, *, B, N, T, in_dim, out_dim, K, static_supports, generic, seed, horizon=1,
               dropout=0.0, train_steps=1, gcn_bool=True, addaptadj=True, aptinit=None):
    sup_t = [torch.from_numpy(s) for s in static_supports]
    ns = R.load_gwnet(generic, sup_t, n_counties=N)
    g = ns['gwnet']('cpu', num_nodes=N, dropout=dropout, supports=sup_t, in_dim=in_dim,
                    out_dim=out_dim, horizon=horizon, kernel_size=K, gcn_bool=gcn_bool, addaptadj=addaptadj,
                    aptinit=aptinit)
    schema = P.gwnet_schema(num_nodes=N, supports_len=len(sup_t) + (1 if (gcn_bool and addaptadj) else 0),
                            in_dim=in_dim, out_dim=out_dim, kernel_size=K, gcn_bool=gcn_bool,
                            addaptadj=addaptadj)
    assert schema_of(g) == [(k, tuple(v)) for k, v in schema.items()], 'schema mismatch'
    P.load_into(g, P.seeded_values(schema, seed))
    g.train()
    if generic:
        x = rand(seed + 1, (B, in_dim, N, T))
    else:
        x = rand(seed + 1, (N, horizon, in_dim))
    x.requires_grad_(True)
    y = g(x)
    tgt = rand(seed + 2, tuple(y.shape))
    loss = F.mse_loss(y, tgt)
    loss.backward()
    d = dict(y=y.detach().numpy(), loss=np.float64(loss.item()), dx=x.grad.numpy(),
             seed=np.int64(seed), receptive_field=np.int64(g.receptive_field))
    d.update(grad_summary(g))
    d.update(buffers(g))
    # eval-mode output with the updated running stats
    g.eval()
    with torch.no_grad():
        d['y_eval'] = g(x.detach()).numpy()
    # adaptive adjacency
    if gcn_bool and addaptadj:
        with torch.no_grad():
            d['adp'] = F.softmax(F.relu(torch.mm(g.nodevec1, g.nodevec2)), dim=1).numpy()
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **d)
    print(name, 'loss', loss.item(), 'y', tuple(y.shape), 'none grads', len(d['none_grads']))


def unet_blocks_case(seed=300):
    ns_g = R.load_gwnet(False, [torch.eye(67)])
    ns = R.load_unet(ns_g['gwnet'])
    d = {}
    # DoubleConv / Down / Up / OutConv at tiny shapes (unet.py:40-92)
    for nm, ctor, shapes in (
            # image widths are multiples of 4 and Up's skip map is exactly twice the upsampled size (what even image
            # sizes give): the shapes the HIP path serves
            ('double_conv', lambda: ns['DoubleConv'](3, 8), [(2, 3, 12, 12)]),
            ('down', lambda: ns['Down'](4, 8), [(3, 4, 16, 16)]),
            ('up', lambda: ns['Up'](16, 8), [(2, 16, 6, 8), (2, 8, 12, 16)]),
            ('outc', lambda: ns['OutConv'](4, 2), [(2, 4, 8, 8)])):
        torch.manual_seed(0)
        m = ctor()
        vals = P.seeded_values({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed)
        P.load_into(m, vals)
        m.train()
        ins = [rand(seed + 10 + i, s).requires_grad_(True) for i, s in enumerate(shapes)]
        y = m(*ins)
        tgt = rand(seed + 20, tuple(y.shape))
        loss = F.mse_loss(y, tgt)
        loss.backward()
        d[nm + '/y'] = y.detach().numpy()
        d[nm + '/loss'] = np.float64(loss.item())
        for i, t in enumerate(ins):
            d[f'{nm}/dx{i}'] = t.grad.numpy()
        for k, prm in m.named_parameters():
            d[f'{nm}/grad/{k}'] = prm.grad.numpy()
        for k, v in buffers(m).items():
            d[f'{nm}/{k}'] = v
        d[nm + '/keys'] = np.array([k for k in m.state_dict().keys()])
    # the composite blocks (unet.py:95-199) at 3 counties x 2 days of 64x64 tiles (the smallest size whose deepest level, 4x4, still has a width divisible by 4): the module globals n_counties /
    # image_dimension the reference reads are set by the loader
    NC, H, S = 3, 2, 64
    ns3 = R.load_unet(ns_g['gwnet'], n_counties=NC, image_dimension=S)
    torch.manual_seed(0)
    con, enc, dec, exp = ns3['Contraction'](2, H), ns3['Encoder'](), ns3['Decoder'](H), ns3['Expansion'](2)
    for j, (nm, m) in enumerate((('contraction', con), ('encoder', enc), ('decoder', dec), ('expansion', exp))):
        P.load_into(m, P.seeded_values({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed + 40 + j))
        m.train()
        if hasattr(m, 'dropout1'):
            m.dropout1.p = 0.0
        d[nm + '/keys'] = np.array(list(m.state_dict().keys()))
    d['composite/seeds'] = np.array([seed + 40 + j for j in range(4)])
    x = rand(seed + 30, (NC, H, 2, S, S)).requires_grad_(True)
    feat = con(x)                                              # (3, 2, 1024) -> (3, 2, 256)
    fms = con.feature_maps
    z = enc(feat)                                              # (3, 2, 1024) -> (3, 2, 256)
    e = dec(z)                                                 # (3, 2, 64, 2, 2)
    y = exp(e, fms)                                            # (3, 2, 2, 32, 32)
    loss = F.mse_loss(y, rand(seed + 31, tuple(y.shape)))
    loss.backward()
    d['composite/feat'] = feat.detach().numpy()
    d['composite/z'] = z.detach().numpy()
    d['composite/e'] = e.detach().numpy()
    d['composite/y'] = y.detach().numpy()
    d['composite/loss'] = np.float64(loss.item())
    d['composite/dx'] = x.grad.numpy()
    for k, fm in enumerate(fms):
        d[f'composite/fm{k}'] = fm.detach().numpy()
    for nm, m in (('contraction', con), ('encoder', enc), ('decoder', dec), ('expansion', exp)):
        for k, prm in m.named_parameters():
            d[f'{nm}/grad/{k}'] = prm.grad.numpy()
        for k, v in buffers(m).items():
            d[f'{nm}/{k}'] = v
    np.savez_compressed(os.path.join(OUT, 'unet_blocks.npz'), **d)
    print('unet_blocks done')


def up_pad_case(seed=340):
    """Up.forward's F.pad branch (unet.py:76-81): skip maps larger than the upsampled map, symmetric and asymmetric
    padding -- the reference's own Up class body, tiny shapes (widths stay multiples of 4, what the conv kernels serve)."""
    ns_g = R.load_gwnet(False, [torch.eye(67)])
    ns = R.load_unet(ns_g['gwnet'])
    d = {}
    for nm, shapes in (('pad_sym', [(2, 16, 6, 8), (2, 8, 16, 20)]), ('pad_asym', [(2, 16, 6, 8), (2, 8, 15, 20)])):
        torch.manual_seed(0)
        m = ns['Up'](16, 8)
        P.load_into(m, P.seeded_values({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed))
        m.train()
        ins = [rand(seed + 10 + i, s).requires_grad_(True) for i, s in enumerate(shapes)]
        y = m(*ins)
        loss = F.mse_loss(y, rand(seed + 20, tuple(y.shape)))
        loss.backward()
        d[nm + '/y'] = y.detach().numpy()
        d[nm + '/loss'] = np.float64(loss.item())
        for i, t in enumerate(ins):
            d[f'{nm}/dx{i}'] = t.grad.numpy()
        for k, prm in m.named_parameters():
            d[f'{nm}/grad/{k}'] = prm.grad.numpy()
        for k, v in buffers(m).items():
            d[f'{nm}/{k}'] = v
    d['seed'] = np.int64(seed)
    np.savez_compressed(os.path.join(OUT, 'unet_up_pad.npz'), **d)
    print('unet_up_pad done')


def modified_unet_case(name, B, H, seed, channels=1, size=128):
    """Modified_UNET fwd + MSE + bwd through the reference's own class bodies (unet.py:201-231); `size` is the
    module global image_dimension the reference's Encoder/Decoder read (unet.py:132-136,156-160), `channels` its
    input_channels/output_channels constructor arguments (BASELINE config 3: 13 x 256 x 256 tiles)."""
    sup = [torch.eye(67)]
    ns_g = R.load_gwnet(False, sup)
    ns = R.load_unet(ns_g['gwnet'], image_dimension=size)
    m = ns['Modified_UNET'](st_gnn='gwnet', horizon=H, input_channels=channels, output_channels=channels)
    schema = P.unet_schema(input_channels=channels, output_channels=channels, image_dimension=size)
    assert schema_of(m) == [(k, tuple(v)) for k, v in schema.items()], 'unet schema mismatch'
    P.load_into(m, P.seeded_values(schema, seed))
    m.st_gnn.dropout = 0.0
    for g in m.st_gnn.gconv:
        g.dropout = 0.0
    m.encoder.dropout1.p = 0.0
    m.decoder.dropout1.p = 0.0
    m.train()
    x = rand(seed + 1, (B, 67, H, channels, size, size))
    tdim = rand(seed + 3, (B, 67, H, 64))
    y = m(x, tdim)
    tgt = rand(seed + 2, tuple(y.shape))
    loss = F.mse_loss(y, tgt)
    loss.backward()
    yn = y.detach().numpy()
    d = dict(loss=np.float64(loss.item()), y_shape=np.array(yn.shape),
             y_sample=yn.reshape(-1)[::997].copy(), y_mean=np.float64(yn.mean()),
             y_sqsum=np.float64((yn.astype(np.float64) ** 2).sum()),
             y_first=yn[0, 0, 0, 0].copy(), y_last=yn[-1, -1, -1, -1].copy(),
             seed=np.int64(seed))
    d.update(grad_summary(m, max_full=1100))
    d.update(buffers(m))
    # the same step in float64 (same class bodies): the yardstick for fp32 rounding noise of the
    # gradients, which are sums over millions of pixels with heavy cancellation
    m64 = ns['Modified_UNET'](st_gnn='gwnet', horizon=H, input_channels=channels, output_channels=channels)
    P.load_into(m64, P.seeded_values(schema, seed))
    m64.st_gnn.dropout = 0.0
    for g in m64.st_gnn.gconv:
        g.dropout = 0.0
    m64.encoder.dropout1.p = 0.0
    m64.decoder.dropout1.p = 0.0
    m64 = m64.double().train()
    m64.st_gnn.supports = [s_.double() for s_ in m64.st_gnn.supports]
    y64 = m64(x.double(), tdim.double())
    loss64 = F.mse_loss(y64, tgt.double())
    loss64.backward()
    d['loss64'] = np.float64(loss64.item())
    for k, prm in m64.named_parameters():
        if prm.grad is None:
            continue
        g = prm.grad.detach().numpy()
        if g.size <= 1100:
            d['grad64/' + k] = g
        else:
            d['gsample64/' + k] = g.reshape(-1)[::max(1, g.size // 2048)][:2048].copy()
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **d)
    print(name, 'loss', loss.item(), 'loss64', loss64.item())


def _param_samples(module, tag, d, every=1):
    """A strided sample (<= 512 values) of every `every`-th parameter tensor, keyed `<tag>/<name>`."""
    for j, (k, v) in enumerate(module.named_parameters()):
        if j % every:
            continue
        a = v.detach().numpy().reshape(-1)
        d[f'{tag}/{k}'] = a[::max(1, a.size // 512)][:512].astype(np.float32).copy()


def _unet_model(ns, schema, seed, H, channels, dtype=torch.float32):
    m = ns['Modified_UNET'](st_gnn='gwnet', horizon=H, input_channels=channels, output_channels=channels)
    P.load_into(m, P.seeded_values(schema, seed))
    m.st_gnn.dropout = 0.0
    for g in m.st_gnn.gconv:
        g.dropout = 0.0
    m.encoder.dropout1.p = 0.0
    m.decoder.dropout1.p = 0.0
    m = m.to(dtype).train()
    m.st_gnn.supports = [s_.to(dtype) for s_ in m.st_gnn.supports]
    return m


def unet_trajectory_case(name, B, H, seed, channels=1, size=128, steps=6, with_f64=True):
    """`steps` steps of torch.optim.Adam(lr=1e-3) (lit.py:59-61) on the reference's own Modified_UNET class bodies
    (unet.py:201-231, training_step's forward + MSE of lit.py:29-33), a fresh seeded batch per step: per-step losses
    and strided samples of the parameters after the last step; the same trajectory in float64 as the yardstick."""
    sup = [torch.eye(67)]
    ns_g = R.load_gwnet(False, sup)
    ns = R.load_unet(ns_g['gwnet'], image_dimension=size)
    schema = P.unet_schema(input_channels=channels, output_channels=channels, image_dimension=size)
    d = dict(seed=np.int64(seed), steps=np.int64(steps), shape=np.array([B, 67, H, channels, size, size]))
    for tag, dt in (('', torch.float32),) + ((('64', torch.float64),) if with_f64 else ()):
        m = _unet_model(ns, schema, seed, H, channels, dt)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        losses = []
        for i in range(steps):
            x = rand(seed + 10 + i, (B, 67, H, channels, size, size)).to(dt)
            tdim = rand(seed + 40 + i, (B, 67, H, 64)).to(dt)
            tgt = rand(seed + 70 + i, (B, 67, H, channels, size, size)).to(dt)
            opt.zero_grad(set_to_none=True)
            loss = F.mse_loss(m(x, tdim), tgt)
            loss.backward()
            opt.step()
            losses.append(float(loss))
            print(name, tag or '32', 'step', i, 'loss', losses[-1], flush=True)
        d['losses' + tag] = np.array(losses, dtype=np.float64)
        _param_samples(m, 'p' + tag, d, every=3)
        if not tag:
            d.update(buffers(m))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **d)


def gwnet_trajectory_case(name, *, B, N, T, in_dim, out_dim, K, static_supports, seed, steps=6):
    """The same for gwnet alone in the shape-generic (B,C,N,T) form (BASELINE config 2 at N=300): Adam(1e-3), dropout 0,
    a fresh seeded batch per step, fp32 and float64."""
    d = dict(seed=np.int64(seed), steps=np.int64(steps))
    for tag, dt in (('', torch.float32), ('64', torch.float64)):
        sup_t = [torch.from_numpy(s).to(dt) for s in static_supports]
        ns = R.load_gwnet(True, sup_t, n_counties=N)
        g = ns['gwnet']('cpu', num_nodes=N, dropout=0.0, supports=sup_t, in_dim=in_dim, out_dim=out_dim, kernel_size=K)
        schema = P.gwnet_schema(num_nodes=N, supports_len=len(sup_t) + 1, in_dim=in_dim, out_dim=out_dim, kernel_size=K)
        P.load_into(g, P.seeded_values(schema, seed))
        g = g.to(dt).train()
        g.supports = sup_t
        opt = torch.optim.Adam(g.parameters(), lr=1e-3)
        losses = []
        for i in range(steps):
            x = rand(seed + 10 + i, (B, in_dim, N, T)).to(dt)
            opt.zero_grad(set_to_none=True)
            y = g(x)
            tgt = rand(seed + 70 + i, tuple(y.shape)).to(dt)
            loss = F.mse_loss(y, tgt)
            loss.backward()
            opt.step()
            losses.append(float(loss))
            print(name, tag or '32', 'step', i, 'loss', losses[-1], flush=True)
        d['losses' + tag] = np.array(losses, dtype=np.float64)
        _param_samples(g, 'p' + tag, d)
        if not tag:
            d.update(buffers(g))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **d)


def unet_conditioned_grad_case(name='modified_unet_H7', seed=430, H=7, channels=1, size=128):
    """A better-conditioned gradient golden for the bf16 mode's model-level check: BatchNorm groups of
    H = 7 images (the reference default horizon) instead of 2, and a structured target (half the input -- an identity-like
    task) instead of noise, so that the gradient reaching the deep stages is not a sum of cancelling noise terms.  fp32 and
    float64 runs of the reference's own class bodies; every parameter's strided gradient sample + norm."""
    sup = [torch.eye(67)]
    ns_g = R.load_gwnet(False, sup)
    ns = R.load_unet(ns_g['gwnet'], image_dimension=size)
    schema = P.unet_schema(input_channels=channels, output_channels=channels, image_dimension=size)
    x = rand(seed + 1, (1, 67, H, channels, size, size))
    tdim = rand(seed + 3, (1, 67, H, 64))
    tgt = 0.5 * x
    d = dict(seed=np.int64(seed))
    for tag, dt in (('', torch.float32), ('64', torch.float64)):
        m = _unet_model(ns, schema, seed, H, channels, dt)
        y = m(x.to(dt), tdim.to(dt))
        loss = F.mse_loss(y, tgt.to(dt))
        loss.backward()
        d['loss' + tag] = np.float64(loss.item())
        if not tag:
            yn = y.detach().numpy()
            d['y_sample'] = yn.reshape(-1)[::997].copy()
            d.update(grad_summary(m, max_full=1100))
        else:
            for k, prm in m.named_parameters():
                if prm.grad is None:
                    continue
                g = prm.grad.detach().numpy()
                d['gnorm64/' + k] = np.float64(np.sqrt((g ** 2).sum()))
                if g.size <= 1100:
                    d['grad64/' + k] = g
                else:
                    d['gsample64/' + k] = g.reshape(-1)[::max(1, g.size // 2048)][:2048].copy()
        print(name, tag or '32', 'loss', loss.item(), flush=True)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **d)


def checkpoint_case(name='gwnet_ckpt', seed=1234):
    """A Lightning-shaped checkpoint of the reference's own gwnet class (lit.py:59-72,187-196): default-initialised
    under torch.manual_seed (so the init values pin the constructor's RNG consumption order), trained 3 steps with
    torch.optim.Adam(lr=1e-3) + CosineAnnealingLR(T_max=10) stepped once; then the 4th step's output, loss and
    updated parameters, which a resumed run must reproduce."""
    N, in_dim, out_dim, K = 20, 2, 12, 2
    A20 = P.knn_graph(20)
    sup_t = [torch.from_numpy(asym_adj(A20)), torch.from_numpy(asym_adj(A20.T))]
    ns = R.load_gwnet(True, sup_t, n_counties=N)
    torch.manual_seed(seed)
    g = ns['gwnet']('cpu', num_nodes=N, dropout=0.0, supports=sup_t, in_dim=in_dim, out_dim=out_dim, kernel_size=K,
                    skip_channels=64, end_channels=128)
    d = {'seed': np.int64(seed)}
    for k, v in g.state_dict().items():
        d['init/' + k] = v.detach().numpy().copy()
    names = [k for k, _ in g.named_parameters()]
    opt = torch.optim.Adam(g.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10)
    g.train()

    def step(i):
        x = rand(seed + 10 + i, (3, in_dim, N, 12))
        tgt = rand(seed + 50 + i, (3, out_dim, N, 1))
        opt.zero_grad(set_to_none=True)
        y = g(x)
        loss = F.mse_loss(y, tgt)
        loss.backward()
        opt.step()
        return y.detach().numpy().copy(), float(loss)

    for i in range(3):
        step(i)
    sched.step()                                     # end of "epoch 0" (interval 'epoch', lit.py:66-71)
    for k, v in g.state_dict().items():
        d['sd/' + k] = v.detach().numpy().copy()
    osd = opt.state_dict()
    d['opt/lr'] = np.float64(osd['param_groups'][0]['lr'])
    d['opt/initial_lr'] = np.float64(osd['param_groups'][0]['initial_lr'])
    d['opt/has_state'] = np.array([int(i in osd['state']) for i in range(len(names))])
    for i, k in enumerate(names):
        if i in osd['state']:
            st = osd['state'][i]
            d['opt/exp_avg/' + k] = st['exp_avg'].numpy().copy()
            d['opt/exp_avg_sq/' + k] = st['exp_avg_sq'].numpy().copy()
            d['opt/step/' + k] = np.float64(float(st['step']))
    d['sched/last_epoch'] = np.int64(sched.state_dict()['last_epoch'])
    y4, l4 = step(3)
    d['y4'] = y4
    d['loss4'] = np.float64(l4)
    for k, v in g.named_parameters():
        d['p4/' + k] = v.detach().numpy().copy() if v.numel() <= 4096 else \
            v.detach().numpy().reshape(-1)[::max(1, v.numel() // 2048)][:2048].copy()
    # the scheduler's values for the first 25 epochs (configure_optimizers, lit.py:61)
    o2 = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
    s2 = torch.optim.lr_scheduler.CosineAnnealingLR(o2, T_max=10)
    lrs = []
    for _ in range(25):
        lrs.append(o2.param_groups[0]['lr'])
        o2.step()
        s2.step()
    d['cosine_lrs'] = np.array(lrs, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **d)
    print(name, 'loss4', l4, 'lr', d['opt/lr'])


def checkpoint_unet_case(name='unet_ckpt', seed=4321, H=2, size=128):
    """Row f2 for the FULL model (lit.py:59-72,187-196): the reference's Modified_UNET, default-initialised under
    torch.manual_seed (pins the constructor's RNG consumption for all 254 tensors), 3 steps of Adam(1e-3) + one
    CosineAnnealingLR(T_max=10) epoch; strided samples of every state_dict tensor at init and after step 3, of Adam's
    exp_avg / exp_avg_sq, the scheduler's state, and the 4th step's loss + parameter samples a resumed run must
    reproduce.  (The whole state is 113 MB: the fixture keeps samples, the test regenerates the state by running the
    same three steps and checks it against these samples before it saves / loads / resumes.)"""
    sup = [torch.eye(67)]
    ns_g = R.load_gwnet(False, sup)
    ns = R.load_unet(ns_g['gwnet'], image_dimension=size)
    torch.manual_seed(seed)
    m = ns['Modified_UNET'](st_gnn='gwnet', horizon=H, input_channels=1, output_channels=1)
    m.st_gnn.dropout = 0.0
    for g in m.st_gnn.gconv:
        g.dropout = 0.0
    m.encoder.dropout1.p = 0.0
    m.decoder.dropout1.p = 0.0
    m.train()

    def samp(t):
        a = t.detach().numpy().reshape(-1)
        return a[::max(1, a.size // 256)][:256].copy()

    d = {'seed': np.int64(seed), 'keys': np.array(list(m.state_dict().keys()))}
    for k, v in m.state_dict().items():
        d['init/' + k] = samp(v.float())
    names = [k for k, _ in m.named_parameters()]
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10)

    def step(i):
        x = rand(seed + 10 + i, (1, 67, H, 1, size, size))
        td = rand(seed + 40 + i, (1, 67, H, 64))
        tgt = rand(seed + 70 + i, (1, 67, H, 1, size, size))
        opt.zero_grad(set_to_none=True)
        loss = F.mse_loss(m(x, td), tgt)
        loss.backward()
        opt.step()
        return float(loss.detach())

    d['losses'] = np.array([step(i) for i in range(3)], dtype=np.float64)
    sched.step()
    for k, v in m.state_dict().items():
        d['sd/' + k] = samp(v.float())
    osd = opt.state_dict()
    d['opt/lr'] = np.float64(osd['param_groups'][0]['lr'])
    d['opt/has_state'] = np.array([int(i in osd['state']) for i in range(len(names))])
    for i, k in enumerate(names):
        if i in osd['state']:
            d['opt/exp_avg/' + k] = samp(osd['state'][i]['exp_avg'])
            d['opt/exp_avg_sq/' + k] = samp(osd['state'][i]['exp_avg_sq'])
    d['sched/last_epoch'] = np.int64(sched.state_dict()['last_epoch'])
    d['loss4'] = np.float64(step(3))
    for k, v in m.named_parameters():
        d['p4/' + k] = samp(v)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **d)
    print(name, 'losses', d['losses'], 'loss4', d['loss4'], 'lr', d['opt/lr'])


def csr_case():
    import pandas as pd
    import scipy.sparse as sp
    A = pd.read_csv(f'{R.REF}/data/graph/adj_mx_fl.csv', index_col=0).values.astype(np.float32)
    c = sp.csr_matrix(A)
    ct = sp.csr_matrix(A.T)
    asym = R.load_asym_adj()
    d = dict(adj=A, rowptr=c.indptr.astype(np.int32), colidx=c.indices.astype(np.int32),
             vals=c.data.astype(np.float32), t_rowptr=ct.indptr.astype(np.int32),
             t_colidx=ct.indices.astype(np.int32), t_vals=ct.data.astype(np.float32),
             asym=np.asarray(asym(A)), asym_t=np.asarray(asym(A.T)))
    # reference load_adj('doubletransition') output (graph_wavenet.py:13-32) == [I_N]
    src_ns = {}
    import ast
    tree = ast.parse(open(f'{R.REF}/models/graph_wavenet.py').read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'load_adj']
    exec(compile(ast.Module(body=fn, type_ignores=[]), 'ref_load_adj', 'exec'),
         dict(pd=pd, np=np), src_ns)
    _, _, adjs = src_ns['load_adj'](f'{R.REF}/data/graph/adj_mx_fl.csv', 'doubletransition')
    d['load_adj0'] = adjs[0]
    # asym_adj on the synthetic k-NN graph used by the bench (N=20 case)
    Ak = P.knn_graph(20)
    d['knn20'] = Ak
    d['knn20_asym'] = np.asarray(asym(Ak))
    np.savez_compressed(os.path.join(OUT, 'adjacency.npz'), **d)
    print('adjacency nnz', c.nnz)


def date2vec_case(seed=500):
    d2v = R.load_date2vec()
    m = d2v.Date2Vec(k=64)
    vals = P.seeded_values({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed)
    P.load_into(m, vals)
    m.eval()
    x = torch.tensor([[0, 0, 0, 2018, 10, 10], [0, 0, 0, 2022, 9, 26], [0, 0, 0, 2023, 8, 30]],
                     dtype=torch.float32)
    with torch.no_grad():
        y = m.encode(x)
    np.savez_compressed(os.path.join(OUT, 'date2vec.npz'), x=x.numpy(), y=y.numpy(),
                        seed=np.int64(seed), keys=np.array(list(m.state_dict().keys())))
    print('date2vec', tuple(y.shape))


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ['gwnet', 'blocks', 'csr', 'd2v', 'unet']
    if 'gwnet' in which:
        # R: reference default, 3-D input through the :189/:255 views, K=1, supports=[I]
        gwnet_case('gwnet_R', B=1, N=67, T=7, in_dim=320, out_dim=256, K=1,
                   static_supports=[np.eye(67, dtype=np.float32)], generic=False, seed=100,
                   horizon=7)
        # C1: BASELINE config 1 (N=20, T=12, Cin=2, B=4, K=2, two static supports + adaptive)
        A20 = P.knn_graph(20)
        gwnet_case('gwnet_C1', B=4, N=20, T=12, in_dim=2, out_dim=12, K=2,
                   static_supports=[asym_adj(A20), asym_adj(A20.T)], generic=True, seed=200)
        # C1b: K=2 with T shorter than the receptive field (left pad path) and odd N
        A37 = P.knn_graph(37, seed=3)
        gwnet_case('gwnet_C1b', B=3, N=37, T=5, in_dim=5, out_dim=3, K=2,
                   static_supports=[asym_adj(A37)], generic=True, seed=210)
        # C1c: T longer than the receptive field (T_final > 1)
        gwnet_case('gwnet_C1c', B=2, N=20, T=16, in_dim=4, out_dim=6, K=2,
                   static_supports=[asym_adj(A20), asym_adj(A20.T)], generic=True, seed=220)
    if 'variants' in which:
        # constructor variants of graph_wavenet.py:101 -- no gcn (residual_convs path, :245), static supports
        # only (addaptadj=False, :242-243), SVD-initialised adaptive embeddings (aptinit, :138-142)
        A20 = P.knn_graph(20)
        sup2 = [asym_adj(A20), asym_adj(A20.T)]
        gwnet_case('gwnet_V_nogcn', B=3, N=20, T=12, in_dim=3, out_dim=4, K=2, static_supports=sup2, generic=True,
                   seed=230, gcn_bool=False)
        gwnet_case('gwnet_V_static', B=3, N=20, T=12, in_dim=3, out_dim=4, K=2, static_supports=sup2, generic=True,
                   seed=240, addaptadj=False)
        gwnet_case('gwnet_V_k1', B=2, N=20, T=7, in_dim=6, out_dim=5, K=1, static_supports=sup2, generic=True,
                   seed=250)
    if 'ckpt_unet' in which:
        checkpoint_unet_case()
    if 'up_pad' in which:
        up_pad_case()
    if 'variants2' in which:
        # more than three supports (graph_wavenet.py:124-134: supports_len = len(supports) + 1): 4 and 5 supports in all
        A20 = P.knn_graph(20)
        B20 = P.knn_graph(20, seed=11)
        sup3 = [asym_adj(A20), asym_adj(A20.T), asym_adj(B20)]
        gwnet_case('gwnet_V_s4', B=2, N=20, T=12, in_dim=3, out_dim=4, K=2, static_supports=sup3, generic=True, seed=260)
        gwnet_case('gwnet_V_s5', B=2, N=20, T=6, in_dim=5, out_dim=2, K=2, static_supports=sup3 + [asym_adj(B20.T)],
                   generic=True, seed=270)
    if 'ckpt' in which:
        checkpoint_case()
    if 'blocks' in which:
        unet_blocks_case()
    if 'csr' in which:
        csr_case()
    if 'd2v' in which:
        date2vec_case()
    if 'unet' in which:
        modified_unet_case('modified_unet_B2H2', B=2, H=2, seed=400)
    if 'traj_gwnet' in which:
        A300 = P.knn_graph(300, seed=2)
        gwnet_trajectory_case('traj_gwnet_C2s', B=4, N=300, T=12, in_dim=32, out_dim=12, K=2,
                              static_supports=[asym_adj(A300), asym_adj(A300.T)], seed=600)
    if 'traj_unet' in which:
        unet_trajectory_case('traj_unet_B1H2', B=1, H=2, seed=610)
    if 'traj_unet_c3' in which:
        unet_trajectory_case('traj_unet_C3', B=1, H=2, seed=620, channels=13, size=256, with_f64=True)
    if 'unet_h7' in which:
        unet_conditioned_grad_case()
    if 'unet_c3' in which:
        # BASELINE config 3: 13-channel 256x256 tiles (FC bottleneck 16384 -> 4096 -> 256 -> 1024 -> 16384)
        modified_unet_case('modified_unet_C3', B=1, H=2, seed=410, channels=13, size=256)
