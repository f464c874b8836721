#!/usr/bin/env python3
"""Per-kernel table of a `rocprofv3 --kernel-trace --stats --output-format csv -d DIR` run: python scripts/kernel_stats_table.py DIR [slices]"""
import csv
import glob
import sys


def main():
    f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
    slices = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"total kernel time {tot/1e6:.2f} ms; per slice {tot/1e3/slices:.1f} us")
    for r in rows[:30]:
        name = r["Name"].replace("hps::", "").replace("void ", "")[:92]
        print(f"{name:92s} n={int(r['Calls']):6d} avg_us={float(r['AverageNs'])/1e3:8.1f} per_slice_us={float(r['TotalDurationNs'])/1e3/slices:8.1f}")


if __name__ == "__main__":
    main()
