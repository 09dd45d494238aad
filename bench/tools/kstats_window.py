"""Per-kernel totals of the LAST create_proof in a trace of bench/tools/create_proof_trace.py: the window between the last two
`ipa_s_table`-free gaps is approximated by taking the final `frac` of the trace's span (default the last 12 %)."""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:58]))
rows.sort()
t_end = rows[-1][1]
win_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
lo = t_end - int(win_ms * 1e6)
d = collections.defaultdict(lambda: [0, 0.0])
busy = 0
for s, e, n in rows:
    if s >= lo:
        d[n][0] += 1; d[n][1] += (e - s) / 1e3; busy += e - s
print(f"window {win_ms} ms, kernel time {busy/1e6:.2f} ms")
for k, (c, t) in sorted(d.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{k:60s} n={c:4d} total {t/1e3:8.3f} ms")
