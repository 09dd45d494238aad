import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"].split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = v[len(v) * skip // 100:]
    print(f"{k:62s} n={len(v):4d} avg {sum(v2)/len(v2):9.1f} us  min {min(v2):9.1f}")
