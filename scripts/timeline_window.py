"""Print the kernels of every queue in a short window of a rocprofv3 --kernel-trace CSV (start, duration, gap to the queue's
previous kernel), window = [t0, t0 + len) ms after the start of the last busy window.  usage: timeline_window.py <csv> <t0_ms> <len_ms>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in rows)
w0 = ev[0][0]; last_end = ev[0][1]
for s, e, n, q in ev:
    if s - last_end > 50e6: w0 = s
    last_end = max(last_end, e)
a = w0 + float(sys.argv[2])*1e6; b = a + float(sys.argv[3])*1e6
short = lambda n: n.replace("void hps::", "").replace("hps::", "")[:34]
prev = {}
for s, e, n, q in ev:
    if s >= a and s < b:
        gap = (s - prev[q])/1e3 if q in prev else 0.0
        print(f"{(s - a)/1e3:9.1f} us  q{q}  {'  '*int(q)*6}{short(n):34s} {(e - s)/1e3:7.1f} us  gap {gap:6.1f}")
    if s >= a - 5e6: prev[q] = e
