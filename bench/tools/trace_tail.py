#!/usr/bin/env python3
"""The last `ms` milliseconds of a rocprofv3 kernel trace, launch by launch: start (us since the window opened), duration (us),
queue, short kernel name -- what a single call looks like on the device (which launches overlap, where the gaps are)."""
import csv
import re
import sys

path, ms = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
end = max(r[1] for r in rows)
t0 = end - int(ms * 1e6)
win = [r for r in rows if r[0] >= t0]
base = win[0][0]
for a, b, q, name in win:
    short = re.sub(r"^void (h2::)?", "", name)
    short = re.sub(r"\(.*$", "", short)
    print(f"{(a - base) / 1e3:9.1f} {(b - a) / 1e3:8.1f}  q{q}  {short}")
print(f"# window {(end - base) / 1e3:.1f} us, {len(win)} launches")
