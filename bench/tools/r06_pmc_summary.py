#!/usr/bin/env python3
"""profiles/r06_pmc_traffic.json from the passes of bench/tools/r06_pmc.sh (rocprofv3 --pmc, one counter set per run, no trace domains,
native workloads).  Same keys as the r04 file bench.py used to read (`msm_accumulate_2^20`, `msm_bucket_sort_2^20`, `ntt_2^20`, `ntt_2^22`),
plus `generic_2^20` (the grouped generic best_multiexp: digits, per-group sorts, the 512-lane accumulates of one call) and `sq` (VALU
instructions and wait shares per kernel).  Counter values are KiB per dispatch; FETCH_SIZE is corrected by the calibration measured in the
SAME session (bench/ubench_fetch.hip: reported / true per access pattern), WRITE_SIZE is exact.

    python bench/tools/r06_pmc_summary.py gpurun_out/r06_pmc > profiles/r06_pmc_traffic.json"""
import collections
import csv
import json
import os
import sys

TRUE_KIB = {"cal_stream16": 32768, "cal_rows128": 32768, "cal_runs16": 16384, "cal_runs4": 4096, "cal_gather64": 262144, "cal_write_stream16": 32768,
            "cal_write_rows128": 32768}


def collect(path):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path):
        return per
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        per[r["Counter_Name"]][(name, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return per


def main():
    d = sys.argv[1]
    get = lambda p, load: collect(os.path.join(d, f"{p}_{load}", "p_counter_collection.csv"))
    out = {"source": "bench/tools/r06_pmc.sh on one MI355X: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY, "
                     "each set in its own run, over build/h2bench commit 20 3 1 1 (registered commits, one stream), ntt 20,22, msm 20 (generic, grouped form) and "
                     "build/ubench/ubench_fetch (calibration); summarised by bench/tools/r06_pmc_summary.py",
           "units": "bytes per launch unless a key says otherwise; FETCH_SIZE x read_correction (below), WRITE_SIZE as reported"}
    # ---- calibration of this session
    cf, cw = get("fetch", "cal")["FETCH_SIZE"], get("write", "cal")["WRITE_SIZE"]
    cal = {}
    for k, true_kib in TRUE_KIB.items():
        src = cw if "write" in k else cf
        vals = [v for (name, _), vs in src.items() if name.startswith(k) for v in vs]
        if vals:
            cal[k] = {"true_KiB": true_kib, "reported_KiB_avg": round(sum(vals) / len(vals), 1), "ratio": round(sum(vals) / len(vals) / true_kib, 4)}
    rr = [v["ratio"] for k, v in cal.items() if "write" not in k and "gather" not in k]
    corr = round(1.0 / (sum(rr) / len(rr)), 3) if rr else 2.0
    out["calibration"] = {"kernels": cal, "read_correction": corr,
                          "what": "bench/ubench_fetch.hip under the same counters: every kernel moves a known number of bytes once; ratio = reported / true.  Streaming, "
                                  "128-byte rows and 64- / 16-byte gathered runs all report half; cal_gather64 (random 64-byte slots of a 1 GiB table) reports what it "
                                  "REQUESTS, i.e. half of the 128-byte lines that move: the same x2"}

    def kb(per, counter, pred, how="avg"):
        vals = [v for (name, grid), vs in per[counter].items() if pred(name, grid) for v in vs]
        if not vals:
            return None
        return (max(vals) if how == "max" else sum(vals) / len(vals)) * 1024

    fc, wc = get("fetch", "commit"), get("write", "commit")
    # ---- the registered commit's accumulate (the roofline kernel): 2^20 + 1 points, 17-bit table
    is_acc = lambda n, g: n.startswith("h2::msm_accumulate<0, false, true, 256>") and g == 131072
    f, w = kb(fc, "FETCH_SIZE", is_acc, "max"), kb(wc, "WRITE_SIZE", is_acc, "max")
    if f:
        out["msm_accumulate_2^20"] = {"algorithmic_bytes": 96 * ((1 << 20) + 1), "fetch_bytes_reported_max": int(f), "fetch_bytes_corrected": int(f * corr),
                                      "write_bytes_max": int(w or 0), "ratio_to_algorithmic": round((f * corr + (w or 0)) / (96 * ((1 << 20) + 1)), 2),
                                      "note": "15 gathered 64-byte points per scalar from the 1 GiB table, each half of a 128-byte line, + 63 MB of sorted entries"}
    sort = {}
    for (name, grid), vs in fc["FETCH_SIZE"].items():
        base = name.split("<")[0].replace("h2::", "")
        if base.startswith(("msm_s1_", "msm_s2_")) and max(vs) > 256:
            wv = wc["WRITE_SIZE"].get((name, grid), [0.0])
            sort[f"{name} grid={grid}"] = {"read_bytes_corrected": int(max(vs) * 1024 * corr), "write_bytes": int(max(wv) * 1024)}
    if sort:
        tot = sum(v["read_bytes_corrected"] + v["write_bytes"] for v in sort.values())
        out["msm_bucket_sort_2^20"] = {"kernels": sort, "total_hbm_bytes_corrected": tot, "algorithmic_bytes_model": 244 << 20, "ratio_to_model": round(tot / (244 << 20), 3)}
    # ---- NTT
    fn, wn = get("fetch", "ntt"), get("write", "ntt")
    for key, pred, algo, plan in (("ntt_2^20", lambda n, g: "ntt_pass9<0, 10" in n and g == 262144, 64 << 20, "two passes of 10 stages"),
                                  ("ntt_2^22", lambda n, g: ("ntt_pass9<0, 8" in n or "ntt_pass9<0, 6" in n) and g == 1048576, 256 << 20, "three passes of 8 + 8 + 6 stages")):
        passes = {}
        for (name, grid), vs in fn["FETCH_SIZE"].items():
            if pred(name, grid):
                wv = wn["WRITE_SIZE"].get((name, grid), [0.0])
                passes[name] = {"read_bytes_corrected": int(sum(vs) / len(vs) * 1024 * corr), "write_bytes": int(sum(wv) / len(wv) * 1024)}
        if passes:
            tot = sum(v["read_bytes_corrected"] + v["write_bytes"] for v in passes.values())
            out[key] = {"algorithmic_bytes": algo, "plan": plan, "passes": passes, "total_hbm_bytes_corrected": tot, "ratio_to_algorithmic": round(tot / algo, 3)}
    # ---- generic best_multiexp, grouped form, 2^20 points (one call = digits + conversion + per-group sorts + per-group accumulates + folds)
    fg, wg = get("fetch", "generic"), get("write", "generic")
    gen = {}
    for (name, grid), vs in sorted(fg["FETCH_SIZE"].items()):
        short = name.replace("h2::", "")
        if not short.startswith(("msm_glv_digits", "msm_d1_", "msm_s2_bins", "msm_accumulate<0, false, true, 512>", "msm_bases_to_m9", "msm_s1_prefix")) or max(vs) < 256:
            continue
        if short.startswith("msm_bases_to_m9") and grid != 1 << 20:
            continue
        wv = wg["WRITE_SIZE"].get((name, grid), [0.0])
        gen[f"{short} grid={grid}"] = {"launches_seen": len(vs), "read_bytes_corrected_avg": int(sum(vs) / len(vs) * 1024 * corr), "write_bytes_avg": int(sum(wv) / len(wv) * 1024)}
    if gen:
        lat = [v for k, v in gen.items() if k.startswith("msm_accumulate") and "grid=126976" in k]
        thr = [v for k, v in gen.items() if k.startswith("msm_accumulate") and "grid=131072" in k]
        out["generic_2^20"] = {"kernels": gen, "algorithmic_bytes_per_call": 96 << 20,
                               "accumulate_latency_form": None if not lat else {
                                   "launches_per_call": 3, "hbm_bytes_per_call": 3 * (lat[0]["read_bytes_corrected_avg"] + lat[0]["write_bytes_avg"]),
                                   "note": "three groups (5 + 2 + 2 slices), 248 workgroups of 512 lanes each; the average over the three launch shapes x 3"},
                               "accumulate_throughput_form": None if not thr else {
                                   "launches_per_call": 1, "hbm_bytes_per_call": thr[0]["read_bytes_corrected_avg"] + thr[0]["write_bytes_avg"],
                                   "note": "one group over all nine slices (the form a call takes when another stream's multiexp is in flight)"},
                               "note": "~16.9 M gathered 64-byte points per call from the 128 MiB per-call array of M9 bases (+ phi): the line is 128 bytes, so the gathers "
                                       "move ~2 x 1.08 GB whatever the table's size; against 96 B per (scalar, base) pair = 100.7 MB algorithmic"}
    # ---- SQ shares
    sq = {}
    for load in ("commit", "ntt", "generic"):
        per = get("sq", load)
        for (name, grid), wcyc in per["SQ_WAVE_CYCLES"].items():
            if not name.startswith("h2::") or sum(wcyc) / len(wcyc) < 5e6:
                continue
            avg = lambda c: (sum(per[c].get((name, grid), [0.0])) / max(len(per[c].get((name, grid), [0.0])), 1))
            wave, busy, valu, wait = avg("SQ_WAVE_CYCLES"), avg("SQ_BUSY_CYCLES"), avg("SQ_INSTS_VALU"), avg("SQ_WAIT_INST_ANY")
            sq[f"{name.replace('h2::', '')} grid={grid} [{load}]"] = {
                "SQ_INSTS_VALU_per_launch": int(valu), "SQ_WAVE_CYCLES": int(wave), "SQ_BUSY_CYCLES": int(busy),
                "valu_insts_per_wave_cycle": round(valu / wave, 4) if wave else None, "wait_inst_any_share_of_wave_cycles": round(wait / wave, 4) if wave else None}
    out["sq"] = {"what": "per launch, averaged over the launches seen; SQ_INSTS_VALU counts wave-level instructions, SQ_WAVE_CYCLES the cycles waves were resident (summed over waves), "
                         "SQ_WAIT_INST_ANY the cycles a wave waited on a counter", "kernels": sq}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
