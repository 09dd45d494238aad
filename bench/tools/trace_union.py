#!/usr/bin/env python3
"""Device time per launch of one kernel from a rocprofv3 kernel trace (`--kernel-trace --output-format csv`): the mean launch
duration, and the UNION of the launch intervals divided by the launch count.  With independent commits issued on several
streams the launches of msm_accumulate overlap on the chip; the union is the time the device spent on the kernel, which is
what bench.py reports as roofline.avg_kernel_ms (h2_profile_read_busy).

    python bench/tools/trace_union.py <kernel_trace.csv> [kernel name prefix] [min duration us]"""
import csv
import json
import sys


def main():
    path = sys.argv[1]
    prefix = sys.argv[2] if len(sys.argv) > 2 else "void h2::msm_accumulate<0, false, true, 256>"
    min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 600.0          # the dense 2^20 launches (skewed / small legs are shorter)
    iv = []
    for r in csv.DictReader(open(path)):
        if r["Kernel_Name"].startswith(prefix):
            a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            if (b - a) / 1e3 >= min_us:
                iv.append((a, b))
    iv.sort()
    groups = [[iv[0]]]
    for a in iv[1:]:                                                     # phases of the run are separated by gaps > 3 ms
        if a[0] - groups[-1][-1][1] > 3e6:
            groups.append([a])
        else:
            groups[-1].append(a)
    out = {"trace": path, "kernel": prefix, "min_duration_us": min_us, "phases": []}
    for g in groups:
        tot, (lo, hi) = 0, g[0]
        for a, b in g[1:]:
            if a > hi:
                tot += hi - lo
                lo, hi = a, b
            else:
                hi = max(hi, b)
        tot += hi - lo
        out["phases"].append({"launches": len(g), "mean_launch_us": round(sum(b - a for a, b in g) / len(g) / 1e3, 1),
                              "union_per_launch_us": round(tot / len(g) / 1e3, 1),
                              "span_per_launch_us": round((g[-1][1] - g[0][0]) / len(g) / 1e3, 1)})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
