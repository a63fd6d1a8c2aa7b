"""Per-kernel summary (calls, total, mean, share) from a rocprofv3 rocpd sqlite database -- the --stats table when the tool
wrote a .db instead of CSVs.   python tools/rocpd_stats.py <results.db> [out.csv]"""
import re
import sqlite3
import sys

db = sys.argv[1]
con = sqlite3.connect(db)
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
scol = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
name_col = "display_name" if "display_name" in scol else "kernel_name"
rows = cur.execute(f"select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start), min(d.start), max(d.end) "
                   f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
t0, t1 = min(r[5] for r in rows), max(r[6] for r in rows)
lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
for n, c, t, mn, mx, *_ in rows:
    short = re.sub(r"\(anonymous namespace\)::", "", n)
    short = short.split("(")[0][:110]
    lines.append(f'"{short}",{c},{t},{t / c:.0f},{100.0 * t / tot:.2f},{mn},{mx}')
lines.append(f'"# sum of kernel durations {tot / 1e6:.1f} ms over a {((t1 - t0) / 1e6):.1f} ms window",,,,,,')
out = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
print(out)
