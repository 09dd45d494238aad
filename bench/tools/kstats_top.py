#!/usr/bin/env python3
"""Top kernels of a `rocprofv3 --kernel-trace --stats --output-format csv` run: kstats_top.py <dir> <title> <out.txt>."""
import csv
import glob
import sys

d, title, out_path = sys.argv[1], sys.argv[2], sys.argv[3]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
lines = [title, f"total kernel time {tot / 1e6:.2f} ms"]
for r in rows[:30]:
    name = r["Name"].split("(")[0].replace("void ", "")[:60]
    lines.append("%-62s calls %6d  total_ms %9.3f  avg_us %9.1f  %5.1f%%" % (name, int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6,
                                                                        float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / tot * 100))
open(out_path, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:26]))
