"""Duration of one kernel launch by launch (averages over bins of <bin> launches) from a rocprofv3 --kernel-trace CSV.
usage: kernel_series.py <csv> <kernel name substring> [bin]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))/1e3 for r in rows]
b = int(sys.argv[3]) if len(sys.argv) > 3 else 64
print(f"{len(d)} launches of {sys.argv[2]}; bins of {b}: " + " ".join(f"{sum(d[i:i+b])/len(d[i:i+b]):.0f}" for i in range(0, len(d), b)))
