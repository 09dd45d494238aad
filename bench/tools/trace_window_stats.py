#!/usr/bin/env python3
"""Per-kernel totals inside the last `window_ms` of a rocprofv3 kernel (+ memory-copy) trace: trace_window_stats.py <dir> [window_ms].
The window ends at the last traced operation, so with a probe that repeats its call the table is the LAST repetition alone
(sums of overlapping launches on several streams can exceed the window; trace_gaps.py gives busy / idle of the same window)."""
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
win_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 16.0
iv = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("h2::", "")))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "(copy) " + r.get("Direction", "")))
t_end = max(e for _, e, _ in iv)
lo = t_end - int(win_ms * 1e6)
iv = [x for x in iv if x[0] >= lo]
tot, cnt, mx = defaultdict(int), defaultdict(int), defaultdict(int)
for s, e, n in iv:
    tot[n] += e - s
    cnt[n] += 1
    mx[n] = max(mx[n], e - s)
span = (t_end - min(s for s, _, _ in iv)) / 1e6
print(f"window {span:.2f} ms, {len(iv)} operations, {sum(tot.values()) / 1e6:.2f} ms summed over streams")
print(f"{'kernel':64s} {'calls':>6s} {'total us':>10s} {'mean us':>9s} {'max us':>9s}")
for n in sorted(tot, key=lambda x: -tot[x]):
    print(f"{n[:64]:64s} {cnt[n]:6d} {tot[n] / 1e3:10.1f} {tot[n] / cnt[n] / 1e3:9.1f} {mx[n] / 1e3:9.1f}")
