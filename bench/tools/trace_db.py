#!/usr/bin/env python3
"""Read a rocprofv3 results .db (the default output of this ROCm): `calls` lists the launches of the N-th call (a call starts at
every launch of START_KERNEL); `stats` prints per-kernel totals.  usage: trace_db.py DB calls START_KERNEL N [COUNT] | stats"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select start, end, queue_id, stream_id, name, vgpr_count, grid_x, workgroup_x, lds_size from kernels order by start").fetchall()


def short(name):
    name = re.sub(r"^void (h2::)?", "", name)
    return re.sub(r"\(.*$", "", name)


mode = sys.argv[2] if len(sys.argv) > 2 else "stats"
if mode == "stats":
    agg = {}
    for a, b, q, s, name, *_ in rows:
        k = short(name)
        t = agg.setdefault(k, [0, 0.0, 1e18])
        t[0] += 1
        t[1] += (b - a) / 1e3
        t[2] = min(t[2], (b - a) / 1e3)
    for k, (n, tot, mn) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k[:70]:70s} n={n:5d} avg {tot / n:9.1f} us  min {mn:9.1f}  total {tot / 1e3:9.2f} ms")
else:
    key, nth = sys.argv[3], int(sys.argv[4])
    count = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    starts = [i for i, r in enumerate(rows) if short(r[4]).startswith(key)]
    i0 = starts[nth]
    i1 = starts[nth + count] if nth + count < len(starts) else len(rows)
    base = rows[i0][0]
    for a, b, q, s, name, vg, gx, wx, lds in rows[i0:i1]:
        print(f"{(a - base) / 1e3:9.1f} {(b - a) / 1e3:8.1f}  q{q} s{s}  {short(name)[:60]:60s} vgpr {vg:3d} grid {gx // max(wx, 1):6d} x {wx:4d} lds {lds}")
    print(f"# {(max(r[1] for r in rows[i0:i1]) - base) / 1e3:.1f} us from the first launch to the last end")
