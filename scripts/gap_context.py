"""Print the kernels around the first few idle gaps of more than <us> microseconds in the last busy window of a rocprofv3
--kernel-trace CSV.  usage: gap_context.py <csv> <min_gap_us> [how many] [skip first n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
w0 = 0; last_end = ev[0][1]
for i, (s, e, n) in enumerate(ev):
    if s - last_end > 50e6: w0 = i
    last_end = max(last_end, e)
ev = ev[w0:]
thr = float(sys.argv[2])*1e3; want = int(sys.argv[3]) if len(sys.argv) > 3 else 3; skip = int(sys.argv[4]) if len(sys.argv) > 4 else 5
short = lambda n: n.replace("void hps::", "").replace("hps::", "")[:60]
found = 0
for i in range(1, len(ev)):
    if ev[i][0] - ev[i - 1][1] > thr:
        found += 1
        if found <= skip: continue
        print(f"--- gap of {(ev[i][0] - ev[i-1][1])/1e3:.1f} us before kernel {i}")
        for k in range(max(0, i - 8), min(len(ev), i + 5)):
            print(f"   {(ev[k][0] - ev[i][0])/1e3:9.1f} us  {short(ev[k][2]):60s} {(ev[k][1] - ev[k][0])/1e3:7.1f} us")
        if found >= skip + want: break
