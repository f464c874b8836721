"""Several stages in flight: HOW the stages' slices lie against one another in time.  From a rocprofv3 --kernel-trace CSV of
a run with L engines on L streams: per queue, a slice starts with its k_deposit_tiled; for every slice of every stage, the
offset (as a fraction of that stage's slice period) to the latest slice start of each other stage -- stages evenly staggered
sit at 1/L, 2/L; stages in phase at 0 -- and how long kernels of the same CLASS (particle kernels: deposit / explicit /
advance; grid kernels: everything else) of two stages were resident together.
usage: stage_phase.py <kernel_trace.csv>"""
import bisect, collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in rows)
w0 = ev[0][0]; last_end = ev[0][1]
for s, e, n, q in ev:
    if s - last_end > 50e6: w0 = s
    last_end = max(last_end, e)
ev = [x for x in ev if x[0] >= w0]
starts = collections.defaultdict(list)
for s, e, n, q in ev:
    if "k_deposit_tiled" in n: starts[q].append(s)
qs = [q for q in starts if len(starts[q]) > 100]
print(f"{len(qs)} stage queues: " + ", ".join(f"{q} ({len(starts[q])} slices)" for q in qs))
for a in qs:
    per = (starts[a][-1] - starts[a][0])/(len(starts[a]) - 1)
    for b in qs:
        if a == b: continue
        hist = collections.Counter(); n = 0
        for t in starts[a][len(starts[a])//8:]:
            i = bisect.bisect_right(starts[b], t) - 1
            if i < 0: continue
            f = ((t - starts[b][i])/per) % 1.0
            hist[int(f*10)] += 1; n += 1
        print(f"  slice starts of queue {a} (period {per/1e3:7.1f} us) after the latest start of queue {b}, in tenths of a period: " +
              " ".join(f"{100*hist[k]/max(n,1):4.0f}" for k in range(10)) + " %")
def cls(n): return "P" if ("k_deposit_tiled" in n or "k_explicit_tiled" in n or "k_advance" in n) else "G"
pts = []
for s, e, n, q in ev:
    if q in qs: pts.append((s, 1, cls(n))); pts.append((e, -1, cls(n)))
pts.sort()
depth = collections.Counter(); prev = pts[0][0]; acc = collections.Counter()
for t, d, c in pts:
    key = "P"*min(depth["P"], 3) + "G"*min(depth["G"], 3)
    acc[key or "-"] += t - prev; prev = t; depth[c] += d
tot = sum(acc.values())
print(f"resident kernels by class over {tot/1e6:.1f} ms (P = particle kernel, G = grid kernel):")
for k, v in acc.most_common(): print(f"  {k:8s} {v/1e6:9.2f} ms {100*v/tot:5.1f} %")
