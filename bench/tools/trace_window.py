#!/usr/bin/env python3
"""Busy / idle split of the LAST `window_ms` of a rocprofv3 kernel trace (`--kernel-trace --output-format csv`): the union of all
kernel intervals, the idle remainder, and the time per kernel name.  Used on bench/tools/opening_probe.py to see how much of an
opening-argument round the device spends waiting for the host.

    python bench/tools/trace_window.py <kernel_trace.csv> <window_ms>"""
import collections
import csv
import json
import sys


def main():
    path, win = sys.argv[1], float(sys.argv[2]) * 1e6
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
    end = max(b for _, b, _ in rows)
    rows = sorted(r for r in rows if r[0] >= end - win)
    busy, (lo, hi) = 0, rows[0][:2]
    for a, b, _ in rows[1:]:
        if a > hi:
            busy += hi - lo
            lo, hi = a, b
        else:
            hi = max(hi, b)
    busy += hi - lo
    span = end - rows[0][0]
    per = collections.Counter()
    cnt = collections.Counter()
    for a, b, name in rows:
        key = name.split("(")[0].replace("void h2::", "")[:60]
        per[key] += b - a
        cnt[key] += 1
    out = {"span_ms": round(span / 1e6, 3), "busy_ms": round(busy / 1e6, 3), "idle_ms": round((span - busy) / 1e6, 3), "launches": len(rows),
           "per_kernel_ms": {k: [round(v / 1e6, 3), cnt[k]] for k, v in per.most_common(25)}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
