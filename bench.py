#!/usr/bin/env python
"""Headline benchmark: Graph-WaveNet training windows/s on BASELINE config 2
(N=3000 nodes, C=32, T=12, kernel_size=2, 2 static CSR supports + dense adaptive adjacency).

  python bench.py --gpus N --steps K --warmup W          (N=1 default)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N>1, one rank per GPU)

A step = forward + MSE loss + backward + gradient all-reduce (N>1, RCCL) + Adam, on a synthetic
batch already resident in HBM.  Prints ONE JSON line (rank 0).  The roofline object is for the
dominant kernel (the dense adaptive-adjacency node-axis product, MFMA bound), timed live with HIP
events on the launching stream inside the timed region; the cpu_baseline object times the CPU oracle
(oracle/, a port of the reference forward) on the host cores on a bounded sample.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_NODES, C_IN, T_IN, OUT_DIM, KSIZE = 3000, 32, 12, 12, 2
# compulsory tensor traffic per window, forward (all-fp32 / all-bf16 figures of the survey);
# fwd+bwd counted as 3x.  The benchmark's throughput mode stores a MIX of fp32 and bf16 tensors: alg_bytes() below
# re-does the App. D accounting tensor by tensor with the element sizes the engine really stores.
SURVEY_BYTES_FWD_PER_WINDOW = {'f32': 768.7e6, 'bf16': 384.4e6}
# UNet conv stack, 13x256x256 tile, forward, per-op in+out accounting (fp32); x3 with backward
UNET_BYTES_FWD_PER_TILE = 29.7e6
UNET_FLOP_FWD_PER_TILE = 0.444e9
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16 MFMA peak (not the 2:1-sparsity figure)
PEAK_HBM_GBPS = 8000.0


def pmc_traffic(dtype, batch):
    """HBM-side bytes per launch of the dense-product kernel from the committed rocprofv3 PMC passes
    (profiles/r02_pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this
    command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane reads on gfx950).
    None when no pass matches this dtype/batch."""
    for f in ('r03b_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json', 'r01_pmc_traffic.json'):      # the latest round's passes first
        try:
            e = json.load(open(os.path.join(ROOT, 'profiles', f))).get(f'{dtype}_b{batch}')
            if e:
                return e['bytes_per_launch']
        except Exception:
            pass
    return None


def alg_bytes(dtype):
    """App. D accounting (materialised tensors cross HBM once per producer and once per consumer, elementwise chains
    fused, concat never materialised, x2 of the static supports never materialised) per window, forward, with the
    element size of every tensor as the engine stores it.  Returns (block, step): `block` = the fused gated-TCN +
    diffusion-GCN block's own HBM-bound part (TCN, static-support hops, mlp + residual + BatchNorm); `step` adds the
    adaptive support's hop tensors (moved by the dense-product kernels), the skip path with the exact crop
    optimisation, start conv, head and the adaptive adjacency."""
    bf = dtype == 'bf16'
    e_h = 4                       # layer input x_in / pre-BN h / BN output: fp32 in both modes
    e_gw = 2 if bf else 4         # gated TCN output g as written: bf16 in the throughput mode (+ the fp32 crop rows of the
                                  # skip path, counted with the skip conv below), fp32 otherwise
    e_gs = 2 if bf else 4         # g as the node-axis products and the mlp read it
    e_x = 2 if bf else 4          # diffusion intermediates x1 / x2
    S_static, S = 2, 3
    touts, tins, t = [], [], 13
    for d in [1, 2] * 4:
        tins.append(t); t -= d * (KSIZE - 1); touts.append(t)
    block = step = 0.0
    for Tin, To in zip(tins, touts):
        b = 32 * Tin * e_h                                   # TCN reads x_in
        b += 32 * To * e_gw                                  # writes g
        b += S_static * 32 * To * (e_gs + e_x)               # static hop 1: read g, write x1
        b += 32 * To * (e_gs + 2 * S * e_x + e_h + e_h)      # mlp: g, (x1 direct + x1 gathered | x1, x2 adaptive), x_in crop; write h
        b += 32 * To * 2 * e_h                               # BatchNorm apply: read h, write x_out
        block += b
        step += b + 32 * To * (e_gs + 3 * e_x)               # adaptive hops: read g, write x1, read x1, write x2
        step += 32 * To * 4                                  # skip conv reads g
    Tf = touts[-1]
    step += 256 * Tf * 4 * 2                                 # skip written once (crop optimisation) and read by the head
    step += (C_IN * T_IN + 32 * 13) * 4                      # start conv in + out
    step += (512 * Tf * 2 + OUT_DIM * Tf) * 4                # head: r1 written + read, y written
    return block * N_NODES, step * N_NODES


def unet_alg_bytes_fwd(cin, size, act_dtype):
    """SURVEY 8(d)'s per-op in+out accounting of the UNet conv stack for ONE tile, forward, at the element sizes the
    engine stores (unet_engine.bf_ok: raw conv outputs and pooled maps of <= 32 channels at >= 64x64 are bf16 in the bf16
    mode, and so is the ConvTranspose2d output of up3 / up4; the network input, the other ConvTranspose2d outputs, the second
    conv output of up1..3, everything below 64x64 and the OutConv result are fp32).  All-fp32 it reproduces the survey's 29.7 MB for a 13x256x256 tile."""
    from multimodal_outage_amd.unet_engine import bf_ok, ENC_CH, DEC_CH
    es = lambda co, s: 2 if bf_ok(act_dtype, co, s, s) else 4
    tot, s = 0, size
    conv = lambda ci_bytes, co, s_, eo: ci_bytes + co * s_ * s_ * eo
    # inc
    e1 = es(4, s)
    tot += conv(cin * s * s * 4, 4, s, e1) + conv(4 * s * s * e1, 4, s, e1)
    skips = [(4, s, e1)]
    c_prev, e_prev = 4, e1
    for ci, co in ENC_CH:                                  # Down: pool (read + write), DoubleConv
        s2 = s // 2
        ep = es(co, s2)
        tot += ci * s * s * e_prev + ci * s2 * s2 * ep
        tot += conv(ci * s2 * s2 * ep, co, s2, ep) + conv(co * s2 * s2 * ep, co, s2, ep)
        s, c_prev, e_prev = s2, co, ep
        skips.append((co, s, ep))
    e_in = 4                                               # decoder fc output (fp32)
    for k, (ci, co) in enumerate(DEC_CH, 1):               # Up: ConvTranspose2d, DoubleConv over [skip, up]
        s2 = 2 * s
        # (the upsampled map is bf16 where the streaming ConvTranspose2d kernels write it: up3, up4 in the bf16 mode)
        eu = 2 if (bf_ok(act_dtype, co, s2, s2) and ci <= 16 and ci // 2 <= 8) else 4
        tot += ci * s * s * e_in + (ci // 2) * s2 * s2 * eu
        sk_c, _, sk_e = skips[4 - k]
        eb = es(co, s2)
        e2 = eb if k == 4 else 4
        tot += conv(sk_c * s2 * s2 * sk_e + (ci // 2) * s2 * s2 * eu, co, s2, eb) + conv(co * s2 * s2 * eb, co, s2, e2)
        s, e_in = s2, e2
    tot += 4 * s * s * e_in + cin * s * s * 4              # OutConv
    return float(tot)


def unet_pmc_traffic(batch, horizon, cin, size):
    """HBM bytes per step of the whole UNet leg (every kernel: conv stack, FC bottleneck and its Adam, Graph WaveNet, loss)
    from the committed FETCH_SIZE / WRITE_SIZE passes of tools/bench_unet.py (tools/unet_pmc_summary.py); None when the
    leg's shape is not the profiled one."""
    if (batch, horizon, cin, size) != (1, 2, 13, 256):
        return None
    for f in ('r03b_unet_c3_pmc_traffic.json', 'r03_unet_c3_pmc_traffic.json', 'r02_unet_c3_pmc_traffic.json'):
        try:
            return json.load(open(os.path.join(ROOT, 'profiles', f)))['bytes_per_step']
        except Exception:
            pass
    return None


def host_cores():
    """Cores this process may really use: cgroup CPU quota if set, else affinity, capped at the
    16-core share of a one-GPU box (oversubscribing the quota makes the CPU leg crawl)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, 16))


def gpu_matched_loss(schema, supports, x, y, n_steps):
    """Matched-loss check of the CPU leg: the HIP path from the same seeded weights on the same windows for the same
    number of Adam steps (dropout 0.3 on both sides, so the masks -- torch's CPU generator there, the engine's counter
    hash here -- differ and the losses agree statistically, not bitwise).  Returns {mode: loss after n_steps}."""
    from oracle import params as P
    from multimodal_outage_amd.models.graph_wavenet import gwnet
    from multimodal_outage_amd.trainer import FlatTrainer
    out = {}
    for mode in ('f32', 'bf16'):
        torch.manual_seed(42)
        m = gwnet('cpu', num_nodes=N_NODES, dropout=0.3, supports=supports, in_dim=C_IN, out_dim=OUT_DIM,
                  kernel_size=KSIZE, blocks=4, layers=2)
        P.load_into(m, P.seeded_values(schema, 42))
        m = m.cuda().train()
        m.dense_dtype = mode
        tr = FlatTrainer(m, lr=1e-3).attach()
        xd, yd = x.cuda(), y.cuda()
        for _ in range(n_steps):
            tr.zero_grad()
            loss = F.mse_loss(m(xd), yd)
            loss.backward()
            tr.allreduce()
            tr.step()
        out[mode] = round(float(loss.detach()), 6)
        del m, tr
    return out


def cpu_baseline(supports, max_steps=5, batch=4, budget_s=25.0):
    """The CPU oracle (a port of graph_wavenet.py:191-254: dense einsum supports, unfused ops) on the
    host cores: fwd + MSE + bwd + Adam on batches of `batch` windows of the same workload; bounded
    to about `budget_s` seconds of CPU work."""
    from oracle import params as P
    from oracle import gwnet_ref
    cores = host_cores()
    torch.set_num_threads(cores)
    schema = P.gwnet_schema(num_nodes=N_NODES, supports_len=3, in_dim=C_IN, out_dim=OUT_DIM, kernel_size=KSIZE)
    p = P.as_param_dict(P.seeded_values(schema, 42))
    leaves = [v for v in p.values() if v.requires_grad]
    opt = torch.optim.Adam(leaves, lr=1e-3)
    sup = [torch.from_numpy(s) for s in supports]
    g = torch.Generator().manual_seed(42)
    x = torch.randn(batch, C_IN, N_NODES, T_IN, generator=g)
    y = torch.randn(batch, OUT_DIM, N_NODES, 1, generator=g)

    def one():
        opt.zero_grad(set_to_none=True)
        out = gwnet_ref.gwnet_forward(p, x, supports=sup, kernel_size=KSIZE, dropout=0.3, training=True)
        loss = F.mse_loss(out, y)
        loss.backward()
        opt.step()
        return float(loss.detach())

    t0 = time.perf_counter()
    last = one()   # warm-up
    warm = time.perf_counter() - t0
    print(f'[bench] cpu_baseline warm-up step {warm:.1f} s on {cores} threads', file=sys.stderr, flush=True)
    steps, t0 = 0, time.perf_counter()
    while steps < max_steps and (steps == 0 or time.perf_counter() - t0 + warm < budget_s):
        last = one()
        steps += 1
        print(f'[bench] cpu_baseline step {steps}: {time.perf_counter() - t0:.1f} s', file=sys.stderr, flush=True)
    dt = time.perf_counter() - t0
    return {"value": round(batch * steps / dt, 4), "unit": "windows/s", "cores": cores, "kind": "port",
            "sample": f"{steps} step(s) of fwd+MSE+bwd+Adam at batch {batch} (N=3000,T=12,C=32,K=2, "
                      f"dropout 0.3) after 1 warm-up step; {dt:.1f} s",
            "loss_after_steps": round(last, 6), "steps_incl_warmup": steps + 1,
            "gpu_loss_same_steps": gpu_matched_loss(schema, supports, x, y, steps + 1)}


def unet_leg(world, dev, steps=10, warmup=12, batch=1, horizon=2, cin=13, size=256, act_dtype='bf16'):
    """Secondary metric of BASELINE.json ("+ UNet tiles/sec"): Modified_UNET training step (forward + MSE/metrics +
    backward + all-reduce + Adam) on synthetic (B,67,H,13,256,256) GOES-style tiles (config 3), batch-sharded
    like the gwnet leg.  Returns the object printed under "unet"."""
    from multimodal_outage_amd.lit import LitModified_UNET
    from multimodal_outage_amd.trainer import FlatTrainer
    torch.manual_seed(42)
    # the lit.py surface itself: LitModified_UNET(st_gnn, horizon, device).training_step(batch) (lit.py:18-43), the batch in
    # the DataLoader's layout (x, y: (B, H, 67, C, S, S); x_time: (B, 67, H, 64), utils.py:101-105) -- the (0,2,1,3,4,5)
    # permute of lit.py:31 is inside the timed step (folded into image offsets, no copy)
    lit = LitModified_UNET('gwnet', horizon, dev, input_channels=cin, output_channels=cin, image_dimension=size)
    m = lit.model.train()
    m.act_dtype = act_dtype      # BASELINE config 3 names bf16 (storage + matrix-pipe arithmetic)
    tr = FlatTrainer(m, eager_adam=True).attach()      # Adam of a module as soon as its gradients are final (trainer.py)
    g = torch.Generator().manual_seed(2000 + int(os.environ.get('RANK', '0')))
    x = torch.randn(batch, horizon, 67, cin, size, size, generator=g).to(dev)
    y = torch.randn(batch, horizon, 67, cin, size, size, generator=g).to(dev)
    td = torch.randn(batch, 67, horizon, 64, generator=g).to(dev)

    def step():
        tr.zero_grad()
        loss = lit.training_step((x, y, td))
        loss.backward()
        tr.allreduce()
        tr.step()
        return loss.detach()       # (the graph behind `loss` holds the step's activations as Function attributes: do not keep it)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    wev = []
    for i in range(warmup):                 # (the caching allocator's pool settles after ~10 steps of this leg -- the gwnet leg
        step()                              #  before it returned its memory to the driver -- under the same two steps of
        wev.append(torch.cuda.Event())      #  launch-thread lead as the timed pass below)
        wev[-1].record()
        if i >= 2:
            wev[i - 2].synchronize()
    gc.collect()                            # (a full collection of the launch thread's heap now, not inside the pass: the first
    sync()                                  #  timed steps of one profile run took 8 / 11 ms of launch time for a 6.2 ms step)
    # ONE timed pass; every step also gets a host timestamp and a HIP event, so that a stall shows where it sits
    # (launch side vs device side, which step) instead of disappearing in a best-of-N
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    host = [0.0] * (steps + 1)
    ms0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(steps):
        loss = step()
        ev[i + 1].record()
        host[i + 1] = time.perf_counter() - t0
        if os.environ.get('MO_BENCH_DEBUG'):
            ms_ = torch.cuda.memory_stats()
            print(f'[bench] unet step {i}: device allocs so far {ms_.get("num_device_alloc", 0)}, reserved '
                  f'{ms_["reserved_bytes.all.current"] >> 20} MB, alloc retries {ms_.get("num_alloc_retries", 0)}', file=sys.stderr)
        if i >= 2:
            # the launch thread is ~2x faster than the GPU here: left alone it runs many steps ahead, blocks freed on the
            # weight-gradient lane are still pending when the next steps allocate, and the caching allocator grows with
            # hipMalloc calls inside the pass (14 in 10 steps).  Two steps of lead keep the GPU fed and the pool stable.
            ev[i - 1].synchronize()
    sync()
    dt = time.perf_counter() - t0
    ms1 = torch.cuda.memory_stats()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    step_gpu = [round(ev[i].elapsed_time(ev[i + 1]), 2) for i in range(steps)]
    step_host = [round((host[i + 1] - host[i]) * 1e3, 2) for i in range(steps)]
    trace = {"step_ms_gpu": step_gpu, "step_ms_host_launch": step_host,
             "device_allocs_in_pass": int(ms1.get('num_device_alloc', 0) - ms0.get('num_device_alloc', 0)),
             "device_frees_in_pass": int(ms1.get('num_device_free', 0) - ms0.get('num_device_free', 0))}
    tiles = batch * 67 * horizon
    tps = world * tiles * steps / dt
    # conv-stack roofline of this leg (29.7 MB fp32 / 0.444 GFLOP per 13x256x256 tile forward, per-op in+out
    # accounting, x3 with backward) over the WHOLE step time (FC bottleneck, the 67-node Graph WaveNet, loss and Adam
    # are inside it); per-kernel durations: profiles/r02_unet_c3_kernel_stats.csv
    scale = (cin * size * size) / (13.0 * 256 * 256) if (cin, size) != (13, 256) else 1.0
    alg = unet_alg_bytes_fwd(cin, size, act_dtype)        # at the element sizes stored (the survey's figure is all-fp32)
    gbs = 3 * alg * tps / world / 1e9
    roof = {"bound": "hbm", "kernel": "UNet conv stack (all kernels of the Modified_UNET step)",
            "achieved": round(gbs, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBPS, 4),
            "algorithmic_MB_per_tile_fwd": round(alg / 1e6, 2),
            "survey_all_fp32_MB_per_tile_fwd": round(unet_alg_bytes_fwd(cin, size, 'f32') / 1e6, 2),
            "vector_TFLOPs": round(3 * UNET_FLOP_FWD_PER_TILE * scale * tps / world / 1e12, 2),
            "traffic": unet_pmc_traffic(batch, horizon, cin, size)}
    return {"metric": "UNet (Modified_UNET) train tiles/sec", "value": round(tps, 1), "unit": "tiles/s", "roofline": roof,
            "ms_per_step": round(dt / steps * 1e3, 2), "tiles_per_step_per_gpu": tiles, "steps": steps, "warmup": warmup,
            "trace": trace,
            "dtype": ("bf16: activations of <= 32 channels at >= 64x64 stored as bf16; their 3x3 convs, data and weight "
                      "gradients on the bf16 matrix pipe with fp32 accumulation; everything else fp32"
                      if act_dtype == 'bf16' else "f32"),
            "data": "synthetic",
            "config": {"workload": f"Modified_UNET fwd+MSE+bwd+Adam on ({batch},67,{horizon},{cin},{size},{size}) tiles",
                       "tile": f"{cin}x{size}x{size}", "counties": 67, "horizon": horizon, "parallelism": f"dp{world}"},
            "loss": round(float(loss.detach()), 5)}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks exactly as the driver does (`python -m
    torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same flags>`)
    as a child process, let rank 0's JSON line pass through on the inherited stdout and return the child's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL across processes needs it on this pool
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print('[bench] launching: ' + ' '.join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def rehearse(args, world, rank):
    """The rank-side control flow of main() without any GPU work: barrier, a timed region of K empty steps, max over ranks
    of the elapsed time, ONE line from rank 0.  Used by the CPU test of the launcher (gloo)."""
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "train windows/sec (gwnet N=3k,T=12)", "value": None, "rehearsal": True,
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 3), "scaling": "weak"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=256, help='windows per GPU per step (weak scaling)')
    ap.add_argument('--dtype', choices=['bf16', 'f32'], default='bf16',
                    help='operand type of the dense adaptive-adjacency products (fp32 accumulate either way)')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend ('nccl' = RCCL; 'gloo' only to rehearse "
                    "the multi-rank control flow on a one-GPU box)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-unet', action='store_true', help='skip the secondary UNet tiles/s leg')
    ap.add_argument('--cpu-steps', type=int, default=5)
    ap.add_argument('--nodes', type=int, default=N_NODES, help='graph size (default: BASELINE config 2; smaller only to '
                    'rehearse the control flow)')
    ap.add_argument('--rehearse', action='store_true', help='control flow only (launcher, rendezvous, barrier, max-over-'
                    'ranks timing, rank-0 line) with no GPU work: runs on a CPU-only host with --backend gloo')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # started plainly (`python bench.py --gpus N`): become the launcher.  Nothing has touched the GPU yet (importing
        # torch does not), and the ranks are CHILD processes -- never a re-exec of a process that initialised HIP.
        sys.exit(launch_ranks(args.gpus))
    globals()['N_NODES'] = args.nodes

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    if args.rehearse:
        if world > 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            dist.init_process_group(args.backend if args.backend != 'nccl' else 'gloo')
        return rehearse(args, world, rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(args.backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device('cuda', local if world > 1 else 0)

    import multimodal_outage_amd._lib as L
    L.load()                      # the HIP extension is mandatory
    from multimodal_outage_amd.models.graph_wavenet import gwnet
    from multimodal_outage_amd.trainer import FlatTrainer
    from multimodal_outage_amd import gwnet_engine as engine
    from multimodal_outage_amd.graphs import knn_graph, asym_adj

    A = knn_graph(N_NODES)
    supports = [asym_adj(A), asym_adj(A.T)]
    torch.manual_seed(42)         # lit.py:14
    model = gwnet('cpu', num_nodes=N_NODES, dropout=0.3, supports=supports, in_dim=C_IN, out_dim=OUT_DIM,
                  kernel_size=KSIZE, blocks=4, layers=2).to(dev).train()
    model.dense_dtype = args.dtype
    trainer = FlatTrainer(model, lr=1e-3)
    model._mo_grad_out = trainer.grad_out()
    model._mo_grad_ready = trainer.ready_callback()     # gradient all-reduce starts inside backward (N > 1)
    B = args.batch
    g = torch.Generator().manual_seed(1000 + rank)
    x = torch.randn(B, C_IN, N_NODES, T_IN, generator=g).to(dev)
    y = torch.randn(B, OUT_DIM, N_NODES, 1, generator=g).to(dev)
    n_out = y.numel()
    sums = torch.empty(4, device=dev)
    dy = torch.empty_like(y)
    ws = torch.empty(L.load().mo_metrics_ws_floats(n_out), device=dev)

    def step():
        out = model(x)
        L.call('mo_mse_metrics', L.ptr(out), L.ptr(y), n_out, L.ptr(sums), L.ptr(dy), L.ptr(ws), L.stream())
        out.backward(dy)
        trainer.allreduce()
        trainer.step()

    out_keep = None

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if os.environ.get('MO_MAIN_HI'):        # experiment: the main chain on a high-priority stream
        hi = torch.cuda.Stream(device=dev, priority=-1)
        hi.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(hi)
    for _ in range(args.warmup):
        step()
    sync()
    engine.PROFILE = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    prof, engine.PROFILE = engine.PROFILE, None
    loss = float(sums[0].item()) / n_out
    # the same kernels without any side-stream overlap (2 extra steps, not part of `value`): every C-ABI launch of the
    # step is bracketed by HIP events on its (single) stream, so that each kernel family's un-contended time is known --
    # the dense products' (`*_serial`) and the fused gated-TCN + diffusion-GCN block's own HBM-bound kernels
    engine.SERIAL = True
    engine.PROFILE = []
    per_call = []
    real_call = L.call

    def timed_call(name, *a):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        real_call(name, *a)
        e1.record()
        per_call.append((name, e0, e1))
    L.call = timed_call
    n_serial = 2
    for _ in range(n_serial):
        step()
    sync()
    L.call = real_call
    prof_serial, engine.PROFILE = engine.PROFILE, None
    engine.SERIAL = False
    # ... and the same per-launch events with the step's real stream structure (dA products and the weight-gradient lane
    # beside the main chain): the block's kernels as they run INSIDE the step, stretched by what shares the chip with them
    per_call_in = []

    def timed_call_in(name, *a):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        real_call(name, *a)
        e1.record()
        per_call_in.append((name, e0, e1))
    L.call = timed_call_in
    for _ in range(n_serial):
        step()
    sync()
    L.call = real_call
    by_name_in = {}
    for name, e0, e1 in per_call_in:
        by_name_in[name] = by_name_in.get(name, 0.0) + e0.elapsed_time(e1) / n_serial
    by_name = {}
    for name, e0, e1 in per_call:
        by_name[name] = by_name.get(name, 0.0) + e0.elapsed_time(e1) / n_serial
    BLOCK = ('mo_tcn_fwd', 'mo_tcn_bwd', 'mo_tcn_pack_weights', 'mo_spmm_blk', 'mo_spmm_blk2', 'mo_spmm_csr', 'mo_gcn_mlp_fwd',
             'mo_gcn_mlp_bwd', 'mo_bn_finalize', 'mo_bn_bwd')
    block_ms = sum(by_name.get(k, 0.0) for k in BLOCK)
    block_ms_in = sum(by_name_in.get(k, 0.0) for k in BLOCK)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    windows_per_s = world * B * args.steps / dt

    # live roofline leg: HIP-event durations of every dense node-axis product in the timed region
    gemm_ms = sum(e0.elapsed_time(e1) for (_, _, e0, e1) in prof)
    gemm_flops = sum(f for (_, f, _, _) in prof)
    n_launch = max(len(prof), 1)
    ach = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    ser_ms = sum(e0.elapsed_time(e1) for (_, _, e0, e1) in prof_serial)
    ser_ach = sum(f for (_, f, _, _) in prof_serial) / (ser_ms * 1e-3) / 1e12 if ser_ms > 0 else 0.0
    peak = PEAK_BF16_MFMA_TFLOPS if args.dtype == 'bf16' else PEAK_F32_MFMA_TFLOPS
    kname = ("gemm_bf16_256_kernel<256x256x32, 4-stage LDS-DMA ring, mfma_f32_16x16x32_bf16>" if args.dtype == 'bf16'
             else "mo_gemm_kernel<128,128,16, mfma_f32_32x32x2_f32>")
    roofline = {"bound": "mfma", "kernel": kname + " dense adaptive-adjacency node-axis product "
                                                   "(forward, data-gradient and dA launches)",
                "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4),
                "launches": len(prof), "avg_launch_ms": round(gemm_ms / n_launch, 4),
                "avg_launch_gflop": round(gemm_flops / n_launch / 1e9, 3),
                "note": "timed region: the data-path products run on the main stream, the two dA accumulations of a "
                        "layer on a second stream beside the main chain (their 144-tile grids leave 112 CUs to it) and "
                        "the weight-gradient kernels on a third; *_serial = same launches with every overlap disabled "
                        "(2 extra steps after the timed region); peak is the spec-sheet dense bf16 figure at 2.4 GHz -- "
                        "under this kernel the chip is power-limited and holds ~1.54 GHz (tools/clock_probe: 2402 MHz "
                        "idle, 1539 MHz beside the GEMM), i.e. 1.6 PFLOP/s at the sustained clock",
                "achieved_serial": round(ser_ach, 3), "frac_serial": round(ser_ach / peak, 4),
                "avg_launch_ms_serial": round(ser_ms / max(len(prof_serial), 1), 4),
                "traffic": pmc_traffic(args.dtype, B)}
    # the fused gated-TCN + diffusion-GCN block (north star: >= 40 % of the HBM roofline): algorithmic bytes of the
    # block's own HBM-bound kernels for the dtype mix actually stored (alg_bytes), over their un-contended HIP-event
    # time in the serial steps; and the whole step priced the same way (the MFMA-bound products are in its denominator)
    step_ms = dt / args.steps * 1e3
    blk_b, stp_b = alg_bytes(args.dtype)
    blk_gb, stp_gb = 3 * blk_b * B / 1e9, 3 * stp_b * B / 1e9
    blk_ach = blk_gb / (block_ms * 1e-3) if block_ms > 0 else 0.0
    hbm_block = {"bound": "hbm",
                 "what": "fused gated-TCN + diffusion-GCN block: its own HBM-bound kernels (TCN fwd/bwd, static-support "
                         "SpMM, mlp + residual + BatchNorm fwd/bwd, their weight gradients) un-contended (serial steps, "
                         "HIP events per launch) vs the App. D compulsory traffic of exactly those tensors at the "
                         "element sizes the engine stores (fwd x3 for fwd+bwd)",
                 "dtype_mix": ("g fp32 + bf16 copy, x1/x2 and their gradients bf16, h / BatchNorm / skip / head fp32"
                               if args.dtype == 'bf16' else "all fp32"),
                 "algorithmic_MB_per_window_fwd": round(blk_b / 1e6, 1),
                 "algorithmic_GB_per_step": round(blk_gb, 3), "kernel_ms_per_step": round(block_ms, 3),
                 "achieved": round(blk_ach, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                 "frac": round(blk_ach / PEAK_HBM_GBPS, 4),
                 "in_step": {"kernel_ms_per_step": round(block_ms_in, 3),
                             "achieved": round(blk_gb / (block_ms_in * 1e-3), 1) if block_ms_in > 0 else 0.0,
                             "frac": round(blk_gb / (block_ms_in * 1e-3) / PEAK_HBM_GBPS, 4) if block_ms_in > 0 else 0.0,
                             "note": "the same launches timed inside steps with the real stream structure (the dA products "
                                     "and the weight-gradient lane run beside them): HIP events around every launch on its "
                                     "own stream, 2 extra steps"},
                 "kernel_ms": {k: round(v, 3) for k, v in sorted(by_name.items(), key=lambda kv: -kv[1])[:14]},
                 "whole_step": {"algorithmic_MB_per_window_fwd": round(stp_b / 1e6, 1),
                                "algorithmic_GB_per_step": round(stp_gb, 3), "ms_per_step": round(step_ms, 3),
                                "achieved": round(stp_gb / (step_ms * 1e-3), 1),
                                "frac": round(stp_gb / (step_ms * 1e-3) / PEAK_HBM_GBPS, 4),
                                "survey_all_fp32_MB": SURVEY_BYTES_FWD_PER_WINDOW['f32'] / 1e6,
                                "survey_all_bf16_MB": SURVEY_BYTES_FWD_PER_WINDOW['bf16'] / 1e6,
                                "note": "the MFMA-bound dense products are inside this step time; the survey's figures "
                                        "are without the skip-crop optimisation and for uniform element sizes"}}

    line = {"metric": "train windows/sec (gwnet N=3k,T=12)", "value": round(windows_per_s, 3),
            "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(step_ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "gwnet fwd+MSE+bwd+Adam on (B,32,3000,12) graph windows, kernel_size=2, "
                                   "blocks=4, layers=2, 2 static k-NN supports (CSR, nnz 17996 each) + dense "
                                   "adaptive adjacency, dropout 0.3",
                       "batch_per_gpu": B, "global_batch": B * world, "nodes": N_NODES, "seq_len": T_IN,
                       "parallelism": f"dp{world}"},
            "loss": round(loss, 6), "roofline": roofline, "roofline_hbm_block": hbm_block}
    if not args.no_unet:
        # free the gwnet leg's tensors first, then the secondary leg (every rank takes part: it all-reduces)
        del model, trainer, x, y, dy, out_keep
        torch.cuda.empty_cache()
        line["unet"] = unet_leg(world, dev)
        # the per-step costs of that leg (FC bottleneck, Adam, the 67-node Graph WaveNet's launch chain) amortise with the
        # batch: the same step at 4 windows per GPU (536 tiles), reported beside the 1-window figure
        try:
            u4 = unet_leg(world, dev, steps=5, warmup=6, batch=4)
            line["unet"]["at_4_windows_per_gpu"] = {"value": u4["value"], "unit": "tiles/s", "ms_per_step": u4["ms_per_step"],
                                                    "tiles_per_step_per_gpu": u4["tiles_per_step_per_gpu"]}
        except Exception as e:                            # (never lose the line to the extra measurement)
            line["unet"]["at_4_windows_per_gpu"] = {"error": repr(e)[:200]}
        # the exact-fp32 mode of the same step (the mode the 1e-4 parity tests run), beside the bf16 headline
        try:
            uf = unet_leg(world, dev, steps=5, warmup=6, act_dtype='f32')
            line["unet"]["f32_mode"] = {"value": uf["value"], "unit": "tiles/s", "ms_per_step": uf["ms_per_step"]}
        except Exception as e:
            line["unet"]["f32_mode"] = {"error": repr(e)[:200]}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            print("[bench] gpu: " + json.dumps({k: line[k] for k in ("value", "ms_per_step", "loss")}), file=sys.stderr, flush=True)
            line["cpu_baseline"] = cpu_baseline(supports, max_steps=args.cpu_steps)
            line["speedup_vs_cpu"] = round(windows_per_s / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
