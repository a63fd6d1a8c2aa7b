"""From a rocprofv3 rocpd database of a BSCLIP_FORCE_DIST=1 run: for every RCCL kernel dispatch, which other kernels were
executing at the same time, and how much of the collective's duration they cover.   python tools/rocpd_overlap.py <db> [out]"""
import re
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
scol = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
dcol = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
name_col = "display_name" if "display_name" in scol else "kernel_name"
qcol = "queue_id" if "queue_id" in dcol else ("stream_id" if "stream_id" in dcol else "0")
rows = cur.execute(f"select s.{name_col}, d.start, d.end, d.{qcol} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
short = lambda n: re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0][:70]
coll = [(n, a, b, q) for n, a, b, q in rows if re.search(r"nccl|rccl", n, re.I)]
other = [(n, a, b, q) for n, a, b, q in rows if not re.search(r"nccl|rccl", n, re.I)]
lines = [f"{len(coll)} RCCL kernel dispatches, {len(other)} other dispatches"]
tot = cov = 0
per = {}
import bisect
starts = [a for _, a, _, _ in other]
for n, a, b, q in coll:
    i = bisect.bisect_left(starts, a - 5_000_000)
    spans = []
    names = set()
    while i < len(other) and other[i][1] < b:
        on, oa, ob, oq = other[i]
        if ob > a:
            spans.append((max(a, oa), min(b, ob)))
            names.add(short(on))
        i += 1
    spans.sort()
    covered, end = 0, a
    for x, y in spans:
        if y > end:
            covered += y - max(x, end)
            end = y
    tot += b - a
    cov += covered
    key = short(n)
    p = per.setdefault(key, [0, 0, 0, set()])
    p[0] += 1
    p[1] += b - a
    p[2] += covered
    p[3] |= names
for k, (c, t, v, names) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    lines.append(f"{k}: {c} dispatches, {t / c / 1e3:.1f} us mean, {100.0 * v / max(t, 1):.1f} % of their time beside compute kernels")
    lines.append("    concurrent with: " + ", ".join(sorted(names))[:600])
lines.append(f"all collectives: {tot / 1e6:.3f} ms total, {100.0 * cov / max(tot, 1):.1f} % overlapped by compute kernels on other streams")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
