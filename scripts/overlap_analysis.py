"""Several stages in flight on one GPU (bench.py --inflight L): how the kernels of the L streams overlap.  From a rocprofv3
--kernel-trace CSV, inside the last window without a pause of more than 50 ms: union busy time, time with 1, 2, 3+ kernels
resident, per-queue busy time, and each kernel's average duration (compare with the one-stage trace).
usage: overlap_analysis.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in rows)
w0 = ev[0][0]; last_end = ev[0][1]
for s, e, n, q in ev:
    if s - last_end > 50e6: w0 = s
    last_end = max(last_end, e)
ev = [x for x in ev if x[0] >= w0]
w1 = max(e for s, e, n, q in ev)
print(f"window {(w1 - w0)/1e6:.2f} ms, {len(ev)} kernels, {len(set(q for *_, q in ev))} queues")
pts = []
for s, e, n, q in ev: pts.append((s, 1)); pts.append((e, -1))
pts.sort()
depth = 0; prev = w0; hist = collections.Counter()
for t, d in pts:
    hist[min(depth, 4)] += t - prev; prev = t; depth += d
tot = w1 - w0
for k in sorted(hist): print(f"  {k}{'+' if k == 4 else ' '} kernels resident: {hist[k]/1e6:8.2f} ms  {100*hist[k]/tot:5.1f} %")
byq = collections.Counter()
for s, e, n, q in ev: byq[q] += e - s
for q, b in byq.most_common(): print(f"  queue {q}: sum of kernel durations {b/1e6:.2f} ms ({100*b/tot:.0f} % of the window)")
dur = collections.Counter(); cnt = collections.Counter()
for s, e, n, q in ev: dur[n[:70]] += e - s; cnt[n[:70]] += 1
for n, d in dur.most_common(14): print(f"  {n:70s} {cnt[n]:6d} calls  {d/1e3/cnt[n]:8.1f} us each  {100*d/tot:5.1f} % of the window")
