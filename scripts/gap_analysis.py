"""Idle time between consecutive kernels of each queue in a rocprofv3 --kernel-trace CSV, inside the last window without
a pause of more than 100 ms (the timed part of a diagnostic run): which kernels start late, and how much of a slice
is spent between kernels.  usage: gap_analysis.py <kernel_trace.csv> [n_slices]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nsl = int(sys.argv[2]) if len(sys.argv) > 2 else 1
byq = collections.defaultdict(list)
for r in rows: byq[r.get("Queue_Id", "0")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
main = max(byq.values(), key=len)
main.sort()
w0 = main[0][0]
for (s0, e0, n0), (s1, e1, n1) in zip(main, main[1:]):
    if s1 - e0 > 100e6: w0 = s1
w1 = main[-1][1]
print(f"window {(w1-w0)/1e6:.2f} ms")
for q, v in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    v = sorted(x for x in v if x[0] >= w0 and x[1] <= w1)
    if not v: continue
    busy = sum(e - s for s, e, _ in v)
    span = v[-1][1] - v[0][0]
    print(f"queue {q}: {len(v)} kernels, span {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms ({busy/1e3/nsl:.1f} us/slice), idle {(span-busy)/1e6:.2f} ms ({(span-busy)/1e3/nsl:.1f} us/slice)")
    gaps = collections.Counter(); cnt = collections.Counter()
    for (s0, e0, n0), (s1, e1, n1) in zip(v, v[1:]):
        g = s1 - e0
        if g > 0: gaps[n1[:60]] += g; cnt[n1[:60]] += 1
    for n, g in gaps.most_common(10): print(f"   before {n:60s} {g/1e3/nsl:8.2f} us/slice  ({cnt[n]} gaps, {g/1e3/cnt[n]:.1f} us each)")
    dur = collections.Counter(); dc = collections.Counter()
    for s, e, n in v: dur[n[:60]] += e - s; dc[n[:60]] += 1
    for n, g in dur.most_common(6): print(f"   kernel {n:60s} {g/1e3/nsl:8.2f} us/slice  ({dc[n]} calls, {g/1e3/dc[n]:.1f} us each)")
