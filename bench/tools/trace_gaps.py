#!/usr/bin/env python3
"""Idle gaps of the GPU in the last `window_ms` of a rocprofv3 kernel (+ memory-copy) trace: trace_gaps.py <dir> [window_ms] [min_gap_us].
Prints busy / idle totals and the largest gaps with the kernel (or copy) on either side -- where a host round trip sits."""
import csv, glob, sys
d = sys.argv[1]
win_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
iv = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:50]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")[:30]))
iv.sort()
t_end = max(e for _, e, _ in iv)
lo = t_end - int(win_ms * 1e6)
iv = [x for x in iv if x[0] >= lo]
gaps, busy_end, last = [], iv[0][1], iv[0][2]
busy = 0
cur_s, cur_e = iv[0][0], iv[0][1]
for s, e, n in iv[1:]:
    if s > cur_e:
        gaps.append((s - cur_e, last, n, (cur_e - lo) / 1e6))
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
        last = n
    else:
        if e > cur_e:
            cur_e, last = e, n
busy += cur_e - cur_s
span = (t_end - iv[0][0]) / 1e6
print(f"window {span:.2f} ms: busy {busy / 1e6:.2f} ms, idle {span - busy / 1e6:.2f} ms in {len(gaps)} gaps; launches {len(iv)}")
big = sorted(gaps, reverse=True)[:40]
print(f"gaps >= {min_gap} us: {sum(1 for g in gaps if g[0] >= min_gap * 1e3)} totalling {sum(g[0] for g in gaps if g[0] >= min_gap * 1e3) / 1e6:.2f} ms")
for g, a, b, at in big:
    if g < min_gap * 1e3:
        break
    print(f"  {g / 1e3:8.1f} us at {at:7.2f} ms   after {a:52s} before {b}")
