"""Per hardware queue of a rocprofv3 --kernel-trace CSV (last window without a 50 ms pause): busy time, idle time in gaps longer
than 20 us, and which kernels follow those gaps (what the stream was waiting to start).  usage: ring_gap_compare.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48], r.get("Queue_Id", "0")) for r in rows)
w0 = ev[0][0]; last_end = ev[0][1]
for s, e, n, q in ev:
    if s - last_end > 50e6: w0 = s
    last_end = max(last_end, e)
ev = [x for x in ev if x[0] >= w0]
w1 = max(e for s, e, n, q in ev)
print(f"window {(w1 - w0)/1e6:.1f} ms")
byq = collections.defaultdict(list)
for x in ev: byq[x[3]].append(x)
for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, n, _ in lst)
    gaps = collections.Counter(); gapt = collections.Counter(); big = 0
    prev_end = lst[0][1]; prev_name = lst[0][2]
    for s, e, n, _ in lst[1:]:
        g = s - prev_end
        if g > 20e3:
            gaps[(prev_name[:30], n[:30])] += 1; gapt[(prev_name[:30], n[:30])] += g; big += g
        prev_end = max(prev_end, e); prev_name = n
    print(f"queue {q}: {len(lst)} kernels, busy {busy/1e6:.1f} ms, idle in gaps > 20 us {big/1e6:.1f} ms")
    for k, t in gapt.most_common(5):
        print(f"     {t/1e6:8.1f} ms in {gaps[k]:6d} gaps ({t/gaps[k]/1e3:7.1f} us each)  after [{k[0]}] before [{k[1]}]")
