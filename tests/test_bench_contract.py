"""bench.py's driver contract, on a real GPU: one JSON line with the required keys (incl. `roofline` and
`cpu_baseline`), and the N > 1 path (torch.distributed.run, one rank per GPU) exercised with two ranks over gloo
on whatever GPUs the box has."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _last_json(out: str):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_single_gpu_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and "2^20" in d["config"]["workload"]
    assert abs(d["ms_per_step"] * 1e-3 * d["value"] * 1e6 - (1 << 20)) / (1 << 20) < 0.02     # value == n / time per step
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and r["kernel"] == "msm_accumulate"
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["bit_exact_vs_gpu"] is True
    assert d["value"] > 10 * c["value"]                                   # north star: >= 10x the host CPU, bit-exact
    assert d["checks"]["split_sum_identity"] is True
    assert d["ntt"]["2^22"]["forward_inverse_roundtrip"]["returns_input"] is True
    assert d["ntt"]["2^20"]["forward_inverse_roundtrip"]["returns_input"] is True
    # round 5: the cold-twiddle cost beside every cached figure, the CPU transform as a median, the PMC constants labelled as constants
    tm = d["ntt"]["2^20"]["twiddle_miss"]
    assert tm["first_call_fresh_omega_ms"] > tm["second_call_same_omega_ms"] > 0
    assert len(d["ntt"]["2^20"]["cpu_baseline"]["runs_ms"]) == 5 and d["ntt"]["2^20"]["cpu_baseline"]["bit_exact_vs_gpu"] is True
    assert "NOT measured in this run" in r["traffic_source"]
    assert d["per_rank"] is None                                          # (N = 1: nothing to compare)


def test_bench_two_ranks_gloo():
    env = dict(os.environ, H2_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                          "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 2 and d["checks"]["split_msm_allgather"] is True and d["checks"]["split_sum_identity"] is True
    # whole-job aggregate: both ranks' scalar-mults over the max-over-ranks time
    assert abs(d["ms_per_step"] * 1e-3 * d["value"] * 1e6 - 2 * (1 << 20)) / (2 << 20) < 0.02
    # BASELINE configs[4] is timed on the N > 1 path: 64 column commits over the ranks + the range-split commit and its exchange
    c5 = d["config5"]
    assert c5["columns_total"] == 64 and c5["columns_per_gpu"] == 32 and c5["columns_ms"] > 0 and c5["columns_first_equals_timed_step"] is True
    assert c5["split_equals_whole"] is True and c5["split_commit_ms"] > 0 and c5["h2d_ms_per_32MiB_column_max_over_ranks"] > 0


def test_bench_eight_ranks_gloo_rehearsal():
    """The driver's 8-GPU launch line rehearsed as far as one GPU allows (H2_BENCH_BACKEND=gloo maps the 8 ranks onto the devices
    the box has): eight processes, eight registered tables, per-rank columns, config 5 with 64 / 8 = 8 columns per rank in one
    batched call, the range-split commit over 8 ranges with its all-gather, ONE JSON line last on stdout."""
    env = dict(os.environ, H2_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
                          "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "5",
                          "--warmup", "1", "--no-cpu-baseline", "--no-create-proof", "--prewarm-ms", "50"],
                         capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    assert out.stdout.strip().splitlines()[-1].startswith("{")            # the line is the LAST thing on stdout
    d = _last_json(out.stdout)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 8 and d["scaling"] == "weak"
    assert d["checks"]["split_msm_allgather"] is True and d["checks"]["split_sum_identity"] is True
    assert abs(d["ms_per_step"] * 1e-3 * d["value"] * 1e6 - 8 * (1 << 20)) / (8 << 20) < 0.02      # whole-job aggregate over 8 ranks
    c5 = d["config5"]
    assert c5["columns_total"] == 64 and c5["columns_per_gpu"] == 8 and c5["columns_first_equals_timed_step"] is True
    assert c5["split_equals_whole"] is True and c5["split_commit_ms"] > 0
    assert d["setup"]["bases_register_ms_max_over_ranks"] > 0
    # every rank's own figures (round 5): its K steps alone, its table build, its clock and power under the timed schedule -- the max
    # over ranks is what `ms_per_step` reports, the list is what shows a straggler
    pr = d["per_rank"]
    assert len(pr["ms_per_step_own"]) == 8 and all(0 < x <= d["ms_per_step"] * 1.001 for x in pr["ms_per_step_own"])
    assert len(pr["bases_register_ms"]) == 8 and max(pr["bases_register_ms"]) == d["setup"]["bases_register_ms_max_over_ranks"]
    assert len(pr["sclk_mhz_median_under_timed_schedule"]) == 8 and len(pr["power_w_median_under_timed_schedule"]) == 8


def test_bench_batched_steps():
    """--batch K: K consecutive steps go to h2_commit_batch_device as one column-batched call; a step is still one full commit, the line
    says so (`config.columns_per_call`, `roofline.columns_per_launch`) and the first output still equals the split-and-sum identity."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--batch", "3",
                          "--no-cpu-baseline", "--no-create-proof", "--prewarm-ms", "20"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["config"]["columns_per_call"] == 3 and d["steps"] == 6
    assert abs(d["roofline"]["columns_per_launch"] - 3.0) < 1e-6 and d["roofline"]["launches"] == 2
    assert abs(d["ms_per_step"] * 1e-3 * d["value"] * 1e6 - (1 << 20)) / (1 << 20) < 0.02
    assert d["checks"]["split_sum_identity"] is True
