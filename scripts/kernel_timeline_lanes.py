#!/usr/bin/env python3
"""Lanes view of a rocprofv3 --kernel-trace run with several engines on one GPU (bench.py --inflight L): one column per
hardware queue, and per queue the busy fraction and the idle time ahead of each slice's first kernel (k_shift_zero).

    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python bench.py --cpu-slices 0 --phase-window 0
    python scripts/kernel_timeline_lanes.py /tmp/kt [rows]
"""
import collections
import csv
import glob
import sys

TAGS = (('k_advance', 'PUSH'), ('k_explicit', 'EXPL'), ('k_deposit', 'DEPO'), ('k_dst', 'POIS'), ('k_smooth', 'mg'), ('k_lower', 'mgL'),
        ('k_hierarchy', 'grad'), ('k_shift', 'shft'), ('k_post', 'post'))


def tag(name):
    for k, t in TAGS:
        if name.startswith(k):
            return t
    return name[:8]


def main():
    f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('hps::', '').replace('void ', ''), r.get('Queue_Id', ''))
                for r in csv.DictReader(open(f)))
    # the stretch where more than one queue launches k_shift_zero: the engines in flight
    shifts = [(s, q) for s, e, n, q in ev if n.startswith('k_shift_zero')]
    qs_all = collections.Counter(q for _, q in shifts)
    lanes = sorted(q for q, c in qs_all.items() if c > 8)
    multi = [s for s, q in shifts if q != lanes[0]] if len(lanes) > 1 else [s for s, q in shifts]
    t_lo, t_hi = multi[len(multi)//4], multi[3*len(multi)//4]
    w = [x for x in ev if t_lo <= x[0] < t_hi and x[3] in lanes]
    print(f"lanes (queues) {lanes}; window {1e-3*(t_hi - t_lo):.0f} us, {len(w)} kernels")
    for q in lanes:
        mine = [x for x in w if x[3] == q]
        busy = sum(e - s for s, e, n, _ in mine)
        starts = [i for i, x in enumerate(mine) if x[2].startswith('k_shift_zero')]
        waits = [mine[i][0] - mine[i - 1][1] for i in starts if i > 0]
        per = [(mine[b][0] - mine[a][0]) for a, b in zip(starts, starts[1:])]
        print(f"  queue {q}: busy {100.0*busy/(t_hi - t_lo):5.1f} %, slices {len(starts)}, period {1e-3*sum(per)/max(len(per), 1):7.1f} us, "
              f"idle ahead of a slice's first kernel {1e-3*sum(waits)/max(len(waits), 1):6.1f} us (max {1e-3*max(waits or [0]):.0f})")
    # concurrency histogram
    pts = sorted([(s, 1) for s, e, n, q in w] + [(e, -1) for s, e, n, q in w])
    hist = collections.Counter(); level = 0; last = pts[0][0]
    for t, d in pts:
        hist[level] += t - last; last = t; level += d
    tot = sum(hist.values())
    print("  kernels running at once: " + ", ".join(f"{k}: {100.0*v/tot:.1f} %" for k, v in sorted(hist.items())))
    t0 = w[0][0]
    for s, e, nm, q in w[:rows]:
        print(f"{(s - t0)/1000:8.1f} {(e - s)/1000:6.1f} " + "            "*lanes.index(q) + tag(nm))


if __name__ == '__main__':
    main()
