"""Where is the GPU idle inside a step?  From a rocprofv3 rocpd database (--kernel-trace): over the last FRAC of the trace (the timed
steps), the union of the kernel intervals, the idle time between them, the time with exactly one / two or more kernels in flight, and
the largest gaps with the kernels on either side.   python tools/rocpd_gaps.py <results.db> [frac=0.5] [top=25] [anchor]
With an anchor (a kernel launched once per step, e.g. loss_prep_kernel) the window is instead the last 4 whole steps: from the 5th-last
launch of the anchor to the last one."""
import re
import sqlite3
import sys

db = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
anchor = sys.argv[4] if len(sys.argv) > 4 else None
con = sqlite3.connect(db)
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
scol = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
name_col = "display_name" if "display_name" in scol else "kernel_name"
rows = cur.execute(f"select d.start, d.end, s.{name_col} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
t0, t1 = rows[0][0], max(r[1] for r in rows)
cut, stop, nsteps = t1 - frac * (t1 - t0), t1, None
if anchor:
    marks = [r[0] for r in rows if anchor in r[2]]
    nsteps = min(4, len(marks) - 1)
    cut, stop = marks[-1 - nsteps], marks[-1]
rows = [r for r in rows if cut <= r[0] < stop]
short = lambda n: re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0][:70]
ev = []
for s, e, _ in rows:
    ev += [(s, 1), (e, -1)]
ev.sort()
depth, last, busy1, busy2, idle = 0, rows[0][0], 0, 0, 0
for t, d in ev:
    span = t - last
    if depth == 0:
        idle += span
    elif depth == 1:
        busy1 += span
    else:
        busy2 += span
    depth += d
    last = t
win = (stop - cut) if anchor else max(r[1] for r in rows) - rows[0][0]
if anchor:
    print(f"{nsteps} steps between launches of {anchor}: {win / nsteps / 1e6:.3f} ms per step")
print(f"window {win / 1e6:.2f} ms, {len(rows)} kernels: idle {idle / 1e6:.3f} ms ({100 * idle / win:.1f} %), one kernel in flight "
      f"{busy1 / 1e6:.2f} ms ({100 * busy1 / win:.1f} %), two or more {busy2 / 1e6:.2f} ms ({100 * busy2 / win:.1f} %)")
# time with exactly one kernel in flight, by that kernel's name, and where in the window it falls (tenths)
solo, where = {}, [0] * 10
live = {}
ev2 = sorted([(s_, 0, i) for i, (s_, e_, _) in enumerate(rows)] + [(e_, -1, i) for i, (s_, e_, _) in enumerate(rows)])
last = ev2[0][0]
for t, kind, i in ev2:
    if len(live) == 1 and t > last:
        n = short(rows[next(iter(live))][2])
        solo[n] = solo.get(n, 0) + (t - last)
        where[min(9, int(10 * (last - cut) / max(1, stop - cut)))] += t - last
    if kind == 0:
        live[i] = 1
    else:
        live.pop(i, None)
    last = t
print("one kernel in flight, by kernel:")
for n, t in sorted(solo.items(), key=lambda kv: -kv[1])[:top]:
    print(f"  {t / 1e6:7.3f} ms   {n}")
print("one kernel in flight, by tenth of the window (ms): " + " ".join(f"{w / 1e6:.2f}" for w in where))
# gaps: walk the intervals in start order keeping the running maximum end
gaps = []
cur_end, cur_name = rows[0][1], rows[0][2]
for s, e, n in rows[1:]:
    if s > cur_end:
        gaps.append((s - cur_end, cur_name, n))
    if e > cur_end:
        cur_end, cur_name = e, n
gaps.sort(reverse=True)
print(f"{len(gaps)} gaps; the largest {top}:")
for g, a, b in gaps[:top]:
    print(f"  {g / 1e3:8.1f} us   after {short(a)}   before {short(b)}")
by = {}
for g, a, b in gaps:
    k = (short(a), short(b))
    by[k] = (by.get(k, (0, 0))[0] + g, by.get(k, (0, 0))[1] + 1)
print("gap time by (kernel before, kernel after):")
for (a, b), (g, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"  {g / 1e6:7.3f} ms in {c:4d} gaps   {a}  ->  {b}")
