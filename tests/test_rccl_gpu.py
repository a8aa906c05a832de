"""RCCL on the box (VERDICT r1 #4/#7a): a world_size-1 `nccl` process group on the one GPU of the test box, with the
trainer's collectives forced on.  A one-rank all-reduce is the identity, so gradients must come out unchanged -- what
this proves is that librccl loads and builds a communicator, that the asynchronous handles started from INSIDE backward
(engine announcements on the weight-gradient lane stream, post-accumulate hooks on the autograd stream) and the
end-of-step pass complete, and that the stream ordering between the producing kernels, RCCL's stream and the fused Adam
kernel is right.  Multi-rank arithmetic is covered under gloo (tests/test_distributed_cpu.py, test_distributed_gpu.py);
the 8-GPU run is the driver's.  comm='abi' runs the same steps through the library's own communicator
(include/mo_hip.h mo_allreduce_*)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    out = {}
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from helpers import rand
        from oracle import params as P
        from oracle import gwnet_ref
        from multimodal_outage_amd.models.graph_wavenet import gwnet
        from multimodal_outage_amd.models.unet import Modified_UNET
        from multimodal_outage_amd.trainer import FlatTrainer
        from multimodal_outage_amd.lit import mse_and_metrics
        out['backend'] = dist.get_backend()
        # ---- Graph WaveNet: engine-driven announcements (mid-backward + end) on the lane stream
        N = 40
        A = P.knn_graph(N, seed=3)
        sup = [gwnet_ref.asym_adj(A), gwnet_ref.asym_adj(A.T)]
        res = {}
        for tag, kw in (('plain', dict()), ('rccl', dict(force_collectives=True)),
                        ('rccl_rs_ag', dict(force_collectives=True, collective='rs_ag')),
                        ('abi', dict(force_collectives=True, comm='abi')),
                        ('abi_rs_ag', dict(force_collectives=True, comm='abi', collective='rs_ag'))):
            m = gwnet('cpu', num_nodes=N, dropout=0.0, supports=sup, in_dim=4, out_dim=3, kernel_size=2)
            schema = P.gwnet_schema(num_nodes=N, supports_len=3, in_dim=4, out_dim=3, kernel_size=2)
            P.load_into(m, P.seeded_values(schema, 77))
            m = m.cuda().train()
            tr = FlatTrainer(m, **kw)
            m._mo_grad_out = tr.grad_out()
            m._mo_grad_ready = tr.ready_callback()
            x = rand(500, (2, 4, N, 12)).cuda()
            tgt = rand(600, (2, 3, N, 1)).cuda()
            announced = []
            for it in range(3):
                tr.zero_grad()
                torch.nn.functional.mse_loss(m(x), tgt).backward()
                announced.append(len(tr._done))
                tr.allreduce()
                if it == 0:
                    g0 = tr.flat_g.clone()
                tr.step()
            torch.cuda.synchronize()
            res[tag] = (g0, tr.flat_p.clone(), announced)
            tr.close()
        for tag in ('rccl', 'rccl_rs_ag', 'abi', 'abi_rs_ag'):
            out[f'gw_{tag}_grad_equal'] = bool(torch.equal(res[tag][0], res['plain'][0]))
            out[f'gw_{tag}_params_equal'] = bool(torch.equal(res[tag][1], res['plain'][1]))
            out[f'gw_{tag}_announced'] = res[tag][2]
        # ---- Modified_UNET: hook-driven buckets (top-level children) for gradients that flow through autograd
        res = {}
        for tag, kw in (('plain', dict()), ('rccl', dict(force_collectives=True))):
            torch.manual_seed(3)
            m = Modified_UNET('gwnet', 2, input_channels=1, output_channels=1)
            m.st_gnn.dropout = 0.0
            m.encoder.dropout1.p = 0.0
            m.decoder.dropout1.p = 0.0
            m = m.cuda().train()
            tr = FlatTrainer(m, **kw)
            x = rand(700, (1, 67, 2, 1, 128, 128)).cuda()
            y = rand(701, (1, 67, 2, 1, 128, 128)).cuda()
            td = rand(702, (1, 67, 2, 64)).cuda()
            announced = []
            for it in range(3):
                tr.zero_grad()
                loss, _, _, _ = mse_and_metrics(m(x, td), y)
                loss.backward()
                announced.append(len(tr._done))
                tr.allreduce()
                tr.step()
            torch.cuda.synchronize()
            res[tag] = (tr.flat_g.clone(), tr.flat_p.clone(), announced, float(loss))
        out['unet_grad_equal'] = bool(torch.equal(res['rccl'][0], res['plain'][0]))
        out['unet_params_equal'] = bool(torch.equal(res['rccl'][1], res['plain'][1]))
        out['unet_announced'] = res['rccl'][2]
        out['unet_loss_finite'] = bool(res['rccl'][3] == res['rccl'][3])
        q.put(out)
    except Exception as e:                      # noqa: BLE001
        import traceback
        q.put({'error': traceback.format_exc()})
        raise
    finally:
        dist.destroy_process_group()


def test_rccl_world1_flat_trainer_gwnet_and_unet():
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(port, q))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=120)
    assert 'error' not in res, res.get('error')
    assert p.exitcode == 0
    assert res['backend'] == 'nccl'
    for tag in ('rccl', 'rccl_rs_ag', 'abi', 'abi_rs_ag'):
        assert res[f'gw_{tag}_grad_equal'] and res[f'gw_{tag}_params_equal'], res
        # engine announcements: the late half mid-backward + the rest at the end of backward, every step
        assert all(a >= 2 for a in res[f'gw_{tag}_announced']), res
    assert res['unet_grad_equal'] and res['unet_params_equal'] and res['unet_loss_finite'], res
    # hook-driven buckets: none in the learning pass, then the five top-level children from inside backward
    assert res['unet_announced'][0] == 0 and all(a >= 4 for a in res['unet_announced'][1:]), res
