#!/usr/bin/env python3
"""Start time and duration of every launch of one kernel from a rocprofv3 kernel trace, in launch order (ms since the first)."""
import csv
import sys

path, prefix = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "void h2::msm_accumulate<0, false, true, 256>"
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(path)) if r["Kernel_Name"].startswith(prefix))
t0 = rows[0][0]
print(" ".join(f"{(a - t0) / 1e6:.1f}:{(b - a) / 1e3:.0f}" for a, b in rows))
