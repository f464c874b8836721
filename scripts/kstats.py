"""Summarise a rocprofv3 --kernel-trace sqlite db: per-kernel count / total / average."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"))
tot = sum(r[2] for r in rows)
nsl = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
print(f"total kernel time {tot:.2f} ms; per slice {tot/nsl*1e3:.1f} us")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print(f"{r[0][:100]:100s} n={r[1]:6d} total_ms={r[2]:9.2f} avg_us={r[3]:8.1f} min={r[4]:7.1f} max={r[5]:8.1f} {100*r[2]/tot:5.1f}% per_slice_us={r[2]/nsl*1e3:7.1f}")
