"""Timeline of ONE UNet training step from a rocprofv3 kernel trace (tools/prof_unet.sh): per-stream busy time, the
main stream's idle gaps and what ends them, and (with --list) every launch in order.
  python tools/unet_timeline.py gpurun_out/r3_unet_ks/run_kernel_trace.csv [--step -2] [--list]"""
import collections
import csv
import re
import sys


def main():
    path = sys.argv[1]
    which = int(sys.argv[sys.argv.index('--step') + 1]) if '--step' in sys.argv else -2
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Stream_Id'],
                 re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')[:60]) for r in rows)
    # a step starts at the first kernel after the last adam_kernel of the previous step's tail
    loss = [i for i, e in enumerate(ev) if 'ut_outc_loss_kernel' in e[3] and ', 0>' in e[3]]
    starts = []
    for i in loss:                       # walk back to the step's first launch: the gap before it holds only Adam / zeroing
        starts.append(i)
    # steps are delimited by consecutive forward-loss launches; the step "which" spans loss[which] .. loss[which+1]
    a, b = loss[which], loss[which + 1]
    win = ev[a:b]
    t0, t1 = win[0][0], win[-1][1]
    print(f'window between two forward-loss launches: {(t1 - t0) / 1e3:.0f} us, {len(win)} launches')
    busy = collections.defaultdict(float)
    for s, e, st, n in win:
        busy[st] += e - s
    for st, v in sorted(busy.items(), key=lambda x: -x[1]):
        print(f'  stream {st}: busy {v / 1e3:8.0f} us')
    main_st = max(busy, key=busy.get)
    last = None
    gaps = collections.Counter()
    tot_gap = 0
    for s, e, st, n in win:
        if st != main_st:
            continue
        if last is not None and s - last > 2000:
            gaps[n] += s - last
            tot_gap += s - last
        last = max(last or 0, e)
    print(f'  main-stream idle (gaps > 2 us): {tot_gap / 1e3:.0f} us; by the kernel that ends the gap:')
    for n, v in gaps.most_common(14):
        print(f'    {v / 1e3:7.0f} us  {n}')
    # union busy (any stream)
    pts = []
    for s, e, _, _ in win:
        pts += [(s, 1), (e, -1)]
    pts.sort()
    lvl, lastt, hist = 0, pts[0][0], collections.Counter()
    for t, d in pts:
        hist[min(lvl, 3)] += t - lastt
        lastt = t
        lvl += d
    print('  kernels in flight -> us:', {k: round(v / 1e3) for k, v in sorted(hist.items())})
    if '--list' in sys.argv:
        for s, e, st, n in win:
            print(f'{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {"M" if st == main_st else "L"} {n}')


if __name__ == '__main__':
    main()
