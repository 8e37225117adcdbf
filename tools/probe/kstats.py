"""Per-kernel totals of a rocprofv3 results database: python kstats.py <dir> <steps>"""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
c = sqlite3.connect(db)
tot = c.execute("select sum(end-start)/1e6 from kernels").fetchone()[0]
print(f"total {tot / steps:.3f} ms/step")
for name, n, ms in c.execute("select name, count(*), sum(end-start)/1e6 from kernels group by name order by 3 desc limit 40"):
    print(f"{ms / steps:8.3f} ms/step {n / steps:7.1f} x {ms / n * 1e3:8.1f} us  {name[:110]}")
