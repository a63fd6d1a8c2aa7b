"""From a rocprofv3 rocpd database of `bench.py` (hipGraph replay, two tower streams): how much of a steady-state step has 0, 1, 2, 3+
kernels executing, and where the idle time sits.   python tools/rocpd_concurrency.py <db> [steps=8]
The step boundaries are the AdamW launches (the last kernel of a step)."""
import re
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
scol = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
name_col = "display_name" if "display_name" in scol else "kernel_name"
rows = cur.execute(f"select s.{name_col}, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
short = lambda n: re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0][:60]
adam = [(a, b) for n, a, b in rows if "adamw" in n.lower()]
# one step may launch several AdamW kernels (one per flat buffer): group launches closer than 1 ms
ends = []
for a, b in adam:
    if ends and a - ends[-1] < 1_000_000:
        ends[-1] = b
    else:
        ends.append(b)
# the roofline probe and the side measurements follow the timed steps: use the longest run of evenly spaced steps
gaps = [ends[i + 1] - ends[i] for i in range(len(ends) - 1)]
med = sorted(gaps)[len(gaps) // 2]
best, run = (0, 0), 0
for i, g in enumerate(gaps):
    run = run + 1 if abs(g - med) < 0.1 * med else 0
    if run > best[0]:
        best = (run, i + 1)
n, last = best
n = min(n, nsteps)
t0, t1 = ends[last - n], ends[last]
ev = []
for name, a, b in rows:
    if b > t0 and a < t1:
        ev.append((max(a, t0), 1, name))
        ev.append((min(b, t1), -1, name))
ev.sort(key=lambda e: (e[0], e[1]))
hist, depth, prev = {}, 0, t0
idle_after = {}
last_ended = None
for t, d, name in ev:
    hist[depth] = hist.get(depth, 0) + (t - prev)
    if depth == 0 and t > prev and last_ended is not None:
        idle_after[short(last_ended)] = idle_after.get(short(last_ended), 0) + (t - prev)
    prev = t
    depth += d
    if d < 0:
        last_ended = name
tot = t1 - t0
print(f"{n} steady-state steps, {tot / n / 1e6:.3f} ms per step (step-end to step-end)")
for k in sorted(hist):
    print(f"  {k} kernel(s) executing: {100.0 * hist[k] / tot:5.1f} %  ({hist[k] / n / 1e6:.3f} ms per step)")
print("  idle time by the kernel that ended before the gap (ms per step):")
for k, v in sorted(idle_after.items(), key=lambda kv: -kv[1])[:12]:
    print(f"    {k:60s} {v / n / 1e6:.3f}")
