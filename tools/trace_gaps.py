#!/usr/bin/env python3
"""Idle time of the GPU in a rocprofv3 --kernel-trace csv: union of the kernel intervals over all streams, the gaps between them
(launch-bound stretches) and, per kernel name, the time it was the ONLY kernel running / ran beside another one.
    python tools/trace_gaps.py <kernel_trace.csv> [--window a:b]   (fractions of the trace, default the last 45 %: the timed iteration)"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    lo, hi = 0.55, 1.0
    if '--window' in sys.argv:
        lo, hi = (float(x) for x in sys.argv[sys.argv.index('--window') + 1].split(':'))
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-60:], r.get('Queue_Id', '')))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    a, b = t0 + (t1 - t0) * lo, t0 + (t1 - t0) * hi
    rows = [r for r in rows if r[0] >= a and r[1] <= b]
    span = rows[-1][1] - rows[0][0]
    ev = []
    for s, e, n, q in rows:
        ev.append((s, 1, n))
        ev.append((e, -1, n))
    ev.sort(key=lambda x: (x[0], x[1]))
    active, last, idle, solo, shared = {}, rows[0][0], 0, defaultdict(int), defaultdict(int)
    gaps = []
    for t, d, n in ev:
        dt = t - last
        if dt > 0:
            k = sum(active.values())
            if k == 0:
                idle += dt
                gaps.append(dt)
            elif k == 1:
                solo[next(x for x, c in active.items() if c)] += dt
            else:
                for x, c in active.items():
                    if c:
                        shared[x] += dt
        active[n] = active.get(n, 0) + d
        last = t
    print(f'window {span / 1e6:.1f} ms, {len(rows)} kernels, idle {idle / 1e6:.1f} ms ({100 * idle / span:.1f} %), queues {len(set(r[3] for r in rows))}')
    gaps.sort()
    if gaps:
        print(f'gaps: n {len(gaps)}, median {gaps[len(gaps) // 2] / 1e3:.1f} us, p90 {gaps[int(len(gaps) * .9)] / 1e3:.1f} us, max {gaps[-1] / 1e3:.1f} us; '
              f'sum of gaps > 20 us: {sum(g for g in gaps if g > 20000) / 1e6:.1f} ms')
    names = sorted(set(solo) | set(shared), key=lambda n: -(solo[n] + shared[n]))
    print(f'{"kernel":60s} {"alone ms":>10s} {"overlapped ms":>14s}')
    for n in names[:24]:
        print(f'{n:60s} {solo[n] / 1e6:10.1f} {shared[n] / 1e6:14.1f}')


if __name__ == '__main__':
    main()
