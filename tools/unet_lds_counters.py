"""Per-kernel LDS counters of a UNet step (tools/unet_lds_counters.sh): bank-conflict cycles against LDS-active cycles and
the share of instruction waits that are LDS waits, largest conflict totals first."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in rows:
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:64]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_BUSY_CYCLES':
        n[k] += 1
print(f'{"kernel":64s} {"launches":>8s} {"conflict Mcyc":>13s} {"LDS active":>10s} {"ratio":>6s} {"LDS wait / any wait":>20s} {"busy Mcyc":>10s}')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]['SQ_LDS_BANK_CONFLICT'])[:28]:
    c, a = v['SQ_LDS_BANK_CONFLICT'], v['SQ_ACTIVE_INST_LDS']
    print(f'{k:64s} {n[k]:8d} {c / 1e6:13.1f} {a / 1e6:10.1f} {c / max(a, 1):6.2f} {v["SQ_WAIT_INST_LDS"] / max(v["SQ_WAIT_INST_ANY"], 1):20.2f} {v["SQ_BUSY_CYCLES"] / 1e6:10.1f}')
