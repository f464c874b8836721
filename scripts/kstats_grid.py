"""Per-(kernel, grid) statistics from a rocprofv3 --kernel-trace sqlite db (per-level view of the multigrid kernels)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
gx = 'grid_size_x' if 'grid_size_x' in cols else [c for c in cols if 'grid' in c][0]
wx = 'workgroup_size_x' if 'workgroup_size_x' in cols else [c for c in cols if 'workgroup' in c][0]
pat = sys.argv[3] if len(sys.argv) > 3 else 'k_smooth'
nsl = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(cur.execute(f"select s.kernel_name, d.{gx}/d.{wx}, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%{pat}%' group by s.kernel_name, d.{gx}/d.{wx} order by 1, 2 desc"))
for r in rows:
    print(f"{r[0][:60]:60s} wgs={r[1]:5d} n/slice={r[2]/nsl:5.2f} avg_us={r[4]:7.1f} min={r[5]:7.1f} per_slice_us={r[3]/nsl*1e3:7.1f}")
