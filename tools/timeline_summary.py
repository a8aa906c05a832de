"""Per-stream busy time and kernel concurrency of the timed gwnet steps, from a rocprofv3 kernel trace
(`rocprofv3 --kernel-trace --output-format csv` of `python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline`).

  python tools/timeline_summary.py gpurun_out/final/ks/run_kernel_trace.csv
"""
import collections
import csv
import re
import sys


def main(path, warmup=3, steps=10):
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Stream_Id'],
                 re.sub(r'\(.*', '', r['Kernel_Name'])[:44]) for r in rows)
    adam = [e for e in ev if 'adam' in e[3]]
    t0, t1 = adam[warmup - 1][1], adam[warmup + steps - 1][1]        # the timed region of the gwnet leg
    win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
    print(f'timed region: {(t1 - t0) / 1e6 / steps:.2f} ms/step')
    busy = collections.defaultdict(float)
    names = collections.defaultdict(collections.Counter)
    for s, e, st, n in win:
        busy[st] += e - s
        names[st][n] += e - s
    for st, v in sorted(busy.items(), key=lambda x: -x[1]):
        top = ', '.join(f'{k} {t / 1e6 / steps:.1f}' for k, t in names[st].most_common(5))
        print(f'stream {st}: busy {v / 1e6 / steps:6.2f} ms/step  ({top})')
    pts = []
    for s, e, _, _ in win:
        pts.append((s, 1))
        pts.append((e, -1))
    pts.sort()
    lvl, last, hist = 0, pts[0][0], collections.Counter()
    for t, d in pts:
        hist[min(lvl, 4)] += t - last
        last = t
        lvl += d
    print('kernels in flight -> ms/step:', {k: round(v / 1e6 / steps, 2) for k, v in sorted(hist.items())})


if __name__ == '__main__':
    main(sys.argv[1])
