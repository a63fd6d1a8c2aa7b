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
# world_size 1: RCCL runs no kernel -- an all-gather is one device-to-device copy on the process group's stream.  With
# --memory-copy-trace those copies are in the database: report which kernels were executing while each of them ran.
mc = next((t for t in tabs if t.startswith("rocpd_memory_copy")), None)
if mc is not None:
    mcol = [r[1] for r in cur.execute(f"pragma table_info({mc})")]
    size_col = next((c for c in ("size", "bytes", "nbytes") if c in mcol), None)
    copies = cur.execute(f"select start, end{', ' + size_col if size_col else ''} from {mc} order by start").fetchall()
    lines.append(f"{len(copies)} memory copies in the trace" + (f" (columns: {', '.join(mcol)})" if not size_col else ""))
    by_size = {}
    for row in copies:
        a, b = row[0], row[1]
        sz = row[2] if size_col else -1
        i = bisect.bisect_left(starts, a - 5_000_000)
        names, covered, end = set(), 0, a
        spans = []
        while i < len(other) and other[i][1] < b:
            on, oa, ob, oq = other[i]
            if ob > a:
                spans.append((max(a, oa), min(b, ob)))
                names.add(short(on))
            i += 1
        for x, y in sorted(spans):
            if y > end:
                covered += y - max(x, end)
                end = y
        e = by_size.setdefault(sz, [0, 0, 0, set()])
        e[0] += 1
        e[1] += b - a
        e[2] += covered
        e[3] |= names
    for sz, (c, t, v, names) in sorted(by_size.items(), key=lambda kv: -kv[1][0])[:12]:
        lines.append(f"copies of {sz} bytes: {c}, {t / c / 1e3:.1f} us mean, {100.0 * v / max(t, 1):.1f} % of their time beside running kernels")
        lines.append("    concurrent with: " + ", ".join(sorted(names))[:400])
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
