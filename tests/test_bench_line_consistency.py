"""The bench line committed as evidence (profiles/r05_final_bench.json: the driver's command on the final tree of the round) against the
contract's arithmetic, recomputed here: value = units / time, roofline.achieved = algorithmic bytes per launch / the kernel's device time
per launch, frac = achieved / peak, the NTT figures from their own times.  A formula that drifts in bench.py shows up as an inconsistent
line the next time the file is refreshed; the judge recomputes the same quantities."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_is_self_consistent():
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_final_bench.json")))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["unit"] == "Mscalar-mults/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["n_gpus"] == 1 and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert isinstance(base, dict)
    n = 1 << 20
    # value: whole-job scalar multiplications per second, inputs resident
    assert abs(d["value"] - n / d["ms_per_step"] / 1e3) / d["value"] < 2e-3
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    algo = 96 * (n + 1) * r["columns_per_launch"]                  # 32 B scalar + 64 B base per pair, the blind's pair included
    assert abs(r["achieved"] - algo / (r["avg_kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 2e-3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["avg_kernel_ms"] <= d["ms_per_step"] * 1.001          # device time per launch (union of the intervals) cannot exceed the step
    assert r["traffic"] > algo                                     # PMC bytes include the gathered lines: never below the algorithmic bytes
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and d["value"] / c["value"] > 10      # north star: >= 10x the host CPU
    for key, log_n in (("2^20", 20), ("2^22", 22)):
        t = d["ntt"][key]
        bf = (1 << (log_n - 1)) * log_n
        assert abs(t["Gbutterflies_per_s"] - bf / (t["ms"] * 1e-3) / 1e9) / t["Gbutterflies_per_s"] < 2e-3
        assert abs(t["algorithmic_GBps"] - 64 * (1 << log_n) / (t["ms"] * 1e-3) / 1e9) / t["algorithmic_GBps"] < 2e-3
        assert t["cpu_baseline"]["bit_exact_vs_gpu"] is True
    rn = d["roofline_ntt"]
    assert abs(rn["frac"] - rn["achieved"] / rn["peak"]) < 1e-4
    assert d["checks"]["split_sum_identity"] is True
    assert d["extra"]["create_proof_simple_example_k20"]["accepted_and_wrong_instance_rejected"] is True
    # round 5, second half: the opening argument as one native call (resident and from host vectors: the same bytes), the generic multiexp as
    # independent calls (the same point from every stream)
    oa = d["extra"]["opening_argument_k20"]
    assert oa["same_proof_bytes"] is True and oa["proof_bytes"] == 32 + 64 * 20 + 64 and 5 < oa["resident_p_poly_ms"] < 40 and 5 < oa["host_vectors_ms"] < 60
    ic = d["generic_best_multiexp"]["independent_calls"]
    assert ic["all_equal_affine"] is True and ic["ms_per_call"] <= d["generic_best_multiexp"]["ms"] * 1.05
    assert abs(ic["Mscalar_mults_per_s"] - (n + 1) / ic["ms_per_call"] / 1e3) / ic["Mscalar_mults_per_s"] < 2e-3
    # round 5: the PMC constants say that they are constants; the cold-twiddle cost stands beside the cached NTT figures
    assert "NOT measured in this run" in r["traffic_source"] and "not measured in this run" in rn["traffic_source"]
    for key in ("2^20", "2^22"):
        tm = d["ntt"][key]["twiddle_miss"]
        assert abs(tm["twiddle_miss_ms"] - (tm["first_call_fresh_omega_ms"] - tm["second_call_same_omega_ms"])) < 1e-3 and tm["twiddle_miss_ms"] > 0
        assert d["ntt"][key]["forward_inverse_roundtrip"]["returns_input"] is True
        assert len(d["ntt"][key]["cpu_baseline"]["runs_ms"]) == 5


def test_eight_rank_rehearsal_line_lists_every_rank():
    """profiles/r05_bench_8rank_rehearsal_one_gpu.json: the driver's N = 8 launch line rehearsed on one GPU (gloo): the whole-job value is
    8 ranks' scalar multiplications over the slowest rank's time, and every rank's own step time, table build, clock and power are listed."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_8rank_rehearsal_one_gpu.json")))
    assert d["n_gpus"] == 8 and d["scaling"] == "weak"
    assert abs(d["value"] - 8 * (1 << 20) / d["ms_per_step"] / 1e3) / d["value"] < 2e-3
    pr = d["per_rank"]
    for k in ("ms_per_step_own", "bases_register_ms", "sclk_mhz_median_under_timed_schedule", "power_w_median_under_timed_schedule"):
        assert len(pr[k]) == 8, k
    assert max(pr["ms_per_step_own"]) <= d["ms_per_step"] * 1.001 and max(pr["bases_register_ms"]) == d["setup"]["bases_register_ms_max_over_ranks"]
    assert d["config5"]["columns_total"] == 64 and d["config5"]["split_equals_whole"] is True
