"""Turn two rocprofv3 counter-collection passes (FETCH_SIZE and WRITE_SIZE, collected SEPARATELY as
MI355X_MICROARCH.md prescribes) of `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-unet` into
  * a per-kernel table  profiles/<tag>_pmc_fetch_write_per_kernel.csv
  * the entry bench.py reads for roofline.traffic in profiles/r02_pmc_traffic.json (key <dtype>_b<batch>).

  python tools/pmc_summary.py --fetch gpurun_out/pmc_fetch/run_counter_collection.csv \
      --write gpurun_out/pmc_write/run_counter_collection.csv --batch 128 --dtype bf16 --tag r01_final_b128

FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is DOUBLED in bytes_per_launch (gfx950 tallies the 128-byte requests
of 16-byte-per-lane reads at 64 B).  Infinity-Cache hits are inside FETCH_SIZE.
"""
import argparse
import collections
import csv
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, KPAD = 3000, 3008
TOUT = [12, 10, 9, 7, 6, 4, 3, 1]


def short(name):
    name = re.sub(r'\(.*$', '', name)
    return name.strip()


def per_kernel(path, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row['Counter_Name'] != counter:
                continue
            k = short(row['Kernel_Name'])
            tot[k] += float(row['Counter_Value'])
            cnt[k] += 1
    return tot, cnt


def algorithmic_gemm_bytes(batch):
    """Compulsory bytes of the dense-product launches of one training step in the throughput mode (bf16 operands,
    bf16-only results except dg and dA): returns (total bytes, launches)."""
    a_bytes = N * KPAD * 2
    tot, n = 0, 0
    for li, t in enumerate(TOUT):
        nj = N * batch * t * 32
        tot += 2 * (a_bytes + 2 * nj + 2 * nj); n += 2                # forward hops: bf16 in, bf16 out
        if li < len(TOUT) - 1:                                        # the last layer's BN output is unused: no backward
            tot += a_bytes + 2 * nj + 4 * nj; n += 1                  # dx1 += A dx2 (bf16 read-modify-write)
            tot += a_bytes + 2 * nj + 8 * nj; n += 1                  # dg  += A dx1 (fp32 read-modify-write)
            tot += 2 * (4 * nj + 2 * N * N * 4); n += 2               # two dA accumulations
    return tot, n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--fetch', required=True)
    ap.add_argument('--write', required=True)
    ap.add_argument('--batch', type=int, required=True)
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--tag', required=True)
    ap.add_argument('--match', default='gemm_bf16', help='substring of the dominant kernel(s)')
    ap.add_argument('--out', default='r02_pmc_traffic.json', help='file under profiles/ that bench.py reads')
    a = ap.parse_args()
    ft, fc = per_kernel(a.fetch, 'FETCH_SIZE')
    wt, wc = per_kernel(a.write, 'WRITE_SIZE')
    rows = []
    for k in ft:
        rows.append((k, fc[k], ft[k] / fc[k], wt.get(k, 0.0) / max(wc.get(k, 1), 1)))
    rows.sort(key=lambda r: -(2 * r[2] + r[3]) * r[1])
    out_csv = os.path.join(ROOT, 'profiles', f'{a.tag}_pmc_fetch_write_per_kernel.csv')
    with open(out_csv, 'w') as f:
        f.write('kernel,launches,FETCH_SIZE_KB_per_launch_raw,WRITE_SIZE_KB_per_launch\n')
        for k, n, fk, wk in rows:
            f.write(f'"{k}",{n},{fk:.1f},{wk:.1f}\n')
    sel = [r for r in rows if a.match in r[0]]
    n = sum(r[1] for r in sel)
    fk = sum(r[2] * r[1] for r in sel) / n
    wk = sum(r[3] * r[1] for r in sel) / n
    alg, nl = algorithmic_gemm_bytes(a.batch)
    entry = {"bytes_per_launch": int((2 * fk + wk) * 1024), "launches": n,
             "FETCH_SIZE_KB_per_launch_raw": round(fk, 1), "WRITE_SIZE_KB_per_launch": round(wk, 1),
             "algorithmic_bytes_per_launch_avg": int(alg / nl),
             "note": f"tools/pmc_summary.py over separate rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE passes of "
                     f"`python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-unet` (batch {a.batch}, {a.dtype}); every "
                     f"{a.match}* launch (incl. the 2 serial roofline steps and, in the throughput mode, the head's 3 bf16 GEMM launches per step, one of them the split-K weight gradient; "
                     f"algorithmic_bytes_per_launch_avg covers the adjacency products); FETCH_SIZE x2 (gfx950 correction); "
                     f"per-kernel table {os.path.basename(out_csv)}"}
    jp = os.path.join(ROOT, 'profiles', a.out)
    d = json.load(open(jp)) if os.path.exists(jp) else {}
    d[f'{a.dtype}_b{a.batch}'] = entry
    json.dump(d, open(jp, 'w'), indent=1)
    print(json.dumps(entry, indent=1))
    print('wrote', out_csv)


if __name__ == '__main__':
    main()
