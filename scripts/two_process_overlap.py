"""Two (or more) PROCESSES on one device (bench.py --gpus N --same-device under rocprofv3 --kernel-trace): do their kernels run side by side?
From the per-process kernel-trace CSVs of one run: inside the window in which all processes launch kernels, the time with kernels of 0, 1, 2, ...
processes resident, and per process its busy time.  usage: two_process_overlap.py <dir with *kernel_trace.csv>"""
import csv, glob, os, sys, collections
files = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))
ev = []
spans = []
for k, f in enumerate(files):
    rows = list(csv.DictReader(open(f)))
    if len(rows) < 1000: continue
    dep = sorted(int(r["Start_Timestamp"]) for r in rows if "k_deposit_tiled" in r["Kernel_Name"])
    spans.append((dep[-min(1000, len(dep))], dep[-1]))       # the last 1000 slices of this process: the timed box
    for r in rows: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
procs = sorted(set(k for *_, k in ev))
print(f"{len(procs)} processes with kernels: " + ", ".join(f"{k}: {sum(1 for e in ev if e[2] == k)} kernels" for k in procs))
# window: from the latest first-kernel-after-warm-up to the earliest last kernel; take the last 60 % of the common span
lo = max(a for a, b in spans); hi = min(b for a, b in spans)
pts = []
for s, e, k in ev:
    if e <= lo or s >= hi: continue
    pts.append((max(s, lo), 1, k)); pts.append((min(e, hi), -1, k))
pts.sort()
depth = collections.Counter(); prev = lo; hist = collections.Counter(); busy = collections.Counter()
for t, d, k in pts:
    active = sum(1 for p in procs if depth[p] > 0)
    hist[active] += t - prev
    for p in procs:
        if depth[p] > 0: busy[p] += t - prev
    prev = t; depth[k] += d
tot = hi - lo
print(f"window {tot/1e6:.1f} ms (every process inside the last 1000 slices of its timed box)")
for a in sorted(hist): print(f"  kernels of {a} process(es) resident: {hist[a]/1e6:8.1f} ms  {100*hist[a]/tot:5.1f} %")
for p in procs: print(f"  process {p}: a kernel resident {100*busy[p]/tot:5.1f} % of the window")
