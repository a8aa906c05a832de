"""HBM traffic of the UNet (config 3) training step from two SEPARATE rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; KB):
sum over every kernel launch of the run / steps -> bytes per step and per tile, per-kernel table.  FETCH_SIZE is doubled
(gfx950 tallies the 128-byte requests of 16-byte-per-lane reads at 64 B; MI355X_MICROARCH.md).

  python tools/unet_pmc_summary.py --fetch <dir>/run_counter_collection.csv --write <dir>/run_counter_collection.csv --steps 4 --tiles 134
"""
import argparse, collections, csv, json, os, re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for row in csv.DictReader(open(path)):
        if row['Counter_Name'] != counter:
            continue
        k = re.sub(r'\(.*$', '', row['Kernel_Name']).strip()
        tot[k] += float(row['Counter_Value']); cnt[k] += 1
    return tot, cnt


ap = argparse.ArgumentParser()
ap.add_argument('--fetch', required=True); ap.add_argument('--write', required=True)
ap.add_argument('--steps', type=int, required=True, help='steps in the profiled run incl. warm-up')
ap.add_argument('--tiles', type=int, default=134)
ap.add_argument('--tag', default='r03_unet_c3')
ap.add_argument('--alg-mb', type=float, default=60.75, help='algorithmic conv-stack MB per tile fwd+bwd at the STORED element sizes '
                '(3 x bench.unet_alg_bytes_fwd: 20.25 MB forward in the bf16 mode since the upsampled maps of up3 / up4 are bf16 -- 21.82 before; the all-fp32 figure of the survey is 89.1)')
a = ap.parse_args()
ft, fc = per_kernel(a.fetch, 'FETCH_SIZE')
wt, wc = per_kernel(a.write, 'WRITE_SIZE')
rows = sorted(((k, fc[k], ft[k], wt.get(k, 0.0)) for k in ft), key=lambda r: -(2 * r[2] + r[3]))
with open(os.path.join(ROOT, 'profiles', f'{a.tag}_pmc_fetch_write_per_kernel.csv'), 'w') as f:
    f.write('kernel,launches_per_step,FETCH_SIZE_MB_per_step_x2,WRITE_SIZE_MB_per_step\n')
    for k, n, fk, wk in rows:
        f.write(f'"{k}",{n / a.steps:.1f},{2 * fk / 1024 / a.steps:.2f},{wk / 1024 / a.steps:.2f}\n')
# kernels of the conv stack proper (3x3 convs, their gradients, activation / BatchNorm / pool / ConvTranspose2d / OutConv
# kernels): what SURVEY 8(d)'s 89 MB per tile (all-fp32, forward x3) accounts for; the rest of the step is the FC bottleneck,
# Adam, the loss and the 67-node Graph WaveNet
CONV_KEYS = ('ut_outc_loss', 'ub_conv', 'ub_wgrad', 'ux_conv', 'uw_wgrad', 'ud_conv', 'ud_wgrad', 'unet_act', 'ut_conv', 'ut_wgrad', 'ut_convt',
             'group_bn', 'uslab', 'nchw_stats', 'nchw_chan', 'conv3x3_flip', 'maxpool', 'mo_gemm_kernel<128, 128, 16, 2, 2, 0, 0, 5',
             'mo_gemm_kernel<64, 128, 16', 'mo_gemm_kernel<64, 64, 32, 2, 2, 1, 1, 0, 2', 'mo_gemm_kernel<32, 128, 32, 1, 4, 1, 0, 4')
conv = sum((2 * ft[k] + wt.get(k, 0.0)) for k in ft if any(c in k for c in CONV_KEYS)) * 1024 / a.steps
total = (2 * sum(ft.values()) + sum(wt.values())) * 1024 / a.steps
entry = {"bytes_per_step": int(total), "MB_per_tile": round(total / a.tiles / 1e6, 1),
         "conv_stack_bytes_per_step": int(conv), "conv_stack_MB_per_tile": round(conv / a.tiles / 1e6, 1),
         "algorithmic_MB_per_tile_fwd_bwd_at_stored_widths": a.alg_mb, "conv_stack_over_algorithmic": round(conv / a.tiles / 1e6 / a.alg_mb, 3), "steps_profiled": a.steps, "tiles_per_step": a.tiles,
         "note": "all kernels of the Modified_UNET step (conv stack, FC bottleneck incl. its 340 MB of weights x (fwd + data "
                 "gradient + weight gradient + Adam), the 67-node Graph WaveNet, loss); FETCH_SIZE x2 + WRITE_SIZE"}
json.dump(entry, open(os.path.join(ROOT, 'profiles', f'{a.tag}_pmc_traffic.json'), 'w'), indent=1)
print(json.dumps(entry, indent=1))
