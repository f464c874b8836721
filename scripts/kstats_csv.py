"""rocprofv3 --kernel-trace sqlite db -> the columns of rocprofv3's kernel_stats.csv (Name, Calls, TotalDurationNs, AverageNs,
Percentage, MinNs, MaxNs, StdDev)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
                        f"avg((d.end-d.start)*(d.end-d.start)) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"))
tot = sum(r[2] for r in rows)
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"')
for r in rows:
    var = max(r[6] - r[3]*r[3], 0.0)
    print(f'"{r[0]}",{r[1]},{r[2]},{r[3]:.6f},{100.0*r[2]/tot:.6f},{r[4]},{r[5]},{var**0.5:.6f}')
