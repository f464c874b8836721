#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace run: kernels of a window in start order with queue ids, and the idle time
between consecutive kernels of the busiest queue.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python bench.py --steps 40 --cpu-slices 0 --inflight 0 --phase-window 0
    python scripts/kernel_timeline.py /tmp/kt [first] [count]
"""
import collections
import csv
import glob
import sys


def main():
    f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
    ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('hps::', ''), r.get('Queue_Id', ''))
                for r in csv.DictReader(open(f)))
    first = int(sys.argv[2]) if len(sys.argv) > 2 else len(ev)*3//4
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 70
    t0 = ev[first][0]
    for s, e, n, q in ev[first:first + count]:
        print(f"{(s - t0)/1000:9.1f} {(e - t0)/1000:9.1f} {(e - s)/1000:7.1f} q{q} {n[:70]}")
    # gaps on the busiest queue over the second half of the run
    half = ev[len(ev)//2:]
    busiest = collections.Counter(q for _, _, _, q in half).most_common(1)[0][0]
    mine = [(s, e, n) for s, e, n, q in half if q == busiest]
    gaps = collections.defaultdict(lambda: [0, 0.0])
    busy = 0.0
    for (s0, e0, n0), (s1, e1, n1) in zip(mine, mine[1:]):
        busy += (e0 - s0)/1000
        g = max(0.0, (s1 - e0)/1000)
        k = (n0.split('(')[0][:40], n1.split('(')[0][:40])
        gaps[k][0] += 1; gaps[k][1] += g
    span = (mine[-1][1] - mine[0][0])/1000
    print(f"\nqueue {busiest}: {len(mine)} kernels over {span:.0f} us, busy {busy:.0f} us ({100*busy/span:.1f} %), idle {span - busy:.0f} us")
    for k, (n, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {g/n:6.1f} us x {n:5d} = {g:8.0f} us   {k[0]}  ->  {k[1]}")


if __name__ == '__main__':
    main()
