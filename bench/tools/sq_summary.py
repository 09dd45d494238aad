#!/usr/bin/env python3
"""Shares of a wave's cycles per kernel from one rocprofv3 --pmc pass over SQ counters (bench/tools/collect_profiles.sh, step 3):
every counter summed over the launches of a kernel and divided by its SQ_WAVE_CYCLES.

    python bench/tools/sq_summary.py gpurun_out/r02_sq_ntt/n_counter_collection.csv gpurun_out/r02_sq_msm/m_counter_collection.csv > profiles/r02_pmc_sq.json"""
import collections
import csv
import json
import sys


def main():
    out = {"source": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS "
                     "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE (one pass, no trace domains) -- python bench.py "
                     "--steps 3 --warmup 1 --prewarm-ms 0 --minimal --no-cpu-baseline --streams 1; every counter divided by SQ_WAVE_CYCLES of "
                     "the same kernel (share of a wave's cycles)", "kernels": {}}
    for path in sys.argv[1:]:
        acc = collections.defaultdict(lambda: collections.Counter())
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        for name, c in acc.items():
            wc = c.get("SQ_WAVE_CYCLES", 0.0)
            if wc < 1e8 or not name.startswith("h2::"):
                continue
            out["kernels"][name] = {k: (round(v / wc, 4) if k != "SQ_WAVE_CYCLES" else v) for k, v in sorted(c.items())}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
