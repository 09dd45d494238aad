#!/usr/bin/env python3
"""Headline benchmark: Pallas MSM M scalar-mults/s (+ Fp NTT G butterflies/s) at k = 20 on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   -> ONE JSON line on rank 0.

  step      one pass of the hot path over one batch = one 2^20-point Pallas `best_multiexp`
            (a column commit, BASELINE.json configs[1]); scalars are seeded synthetic columns already
            resident in HBM, the basis is registered once (Params::g stays on the device).
  N > 1     independent column commits, one stream of columns per GPU (weak scaling, no data-path
            collective -- SURVEY.md section 8e); launched by torch.distributed.run, one rank per GPU.
            After the timed region the ranks also run one range-split MSM and sum the 96-byte
            partials exchanged with a single RCCL all_gather (verification, not timed).
  value     total scalar-mults of all ranks / max-over-ranks wall time of the K steps.
            Each commit carries its blind term (w, r) as Params::commit does (poly/commitment.rs:119-130).
  roofline  dominant kernel = msm_accumulate; achieved = algorithmic bytes per launch (96 B per
            (scalar, base) pair) / its device time per launch measured with HIP events on
            the launching streams inside the timed region (h2_profile_read_busy): launches of different streams
            overlap on the chip, so the per-launch figure is the UNION of the launch intervals / launches (never
            more than ms_per_step); `kernel_ms_isolated` is the same kernel alone on one stream.
  cpu_baseline  the C restatement of the reference's best_multiexp (oracle/, "port") timed on the host
            cores of this box on the same 2^20 workload (median of 10 runs), rank 0 at N = 1 only.
  extra     first-class companions of the headline: the free function best_multiexp(coeffs, bases) on bases
            that are NOT registered (BASELINE configs[1] read literally), the Vesta commit (the curve every
            reference proof uses), the host-pointer h2_commit including the PCIe copy of the scalars, and
            configs[3]: create_proof of examples/simple_example at k = 20.
The oracle is used here only as the timed CPU baseline and to generate inputs -- never in the GPU path.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K_LOG = 20
ALGO_BYTES_PER_PAIR = 96          # 32 B scalar + 64 B affine base, read once
HBM_PEAK_GBS = 8000.0             # MI355X HBM3E spec (MI355X_MICROARCH.md)


def _hwmon_dir(dev_index):
    """amdgpu hwmon directory of the GPU behind cuda:dev_index (freq1_input = shader clock in Hz, power1_input = socket power in uW),
    or None when sysfs does not show one."""
    import glob
    import torch
    cands = [d for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*") if os.path.exists(os.path.join(d, "freq1_input"))]
    if not cands:
        return None
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        want = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        for d in cands:
            if os.path.basename(os.path.realpath(os.path.join(d, "..", ".."))) == want:
                return d
    except Exception:
        pass
    return cands[0] if len(cands) == 1 else None


def _newest_pmc():
    """The newest profiles/r*_pmc_traffic.json (rocprofv3 --pmc passes of bench/tools/r06_pmc.sh and its predecessors) and its name: the HBM
    bytes per launch the roofline objects quote as `traffic` -- constants from that file, never measured in this run."""
    import glob
    try:
        path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]
        return json.load(open(path)), "profiles/" + os.path.basename(path)
    except Exception:
        return None, None


class _ClockSampler:
    """Samples the shader clock and the socket power from sysfs every ~2 ms on a thread while a load runs."""

    def __init__(self, hwmon):
        import threading
        self.hwmon, self.f, self.p, self.stop = hwmon, [], [], False
        self.thread = threading.Thread(target=self._run, daemon=True)

    def _read(self, name):
        with open(os.path.join(self.hwmon, name)) as fh:
            return int(fh.read().strip())

    def _run(self):
        while not self.stop:
            try:
                self.f.append(self._read("freq1_input") / 1e6)
                self.p.append(self._read("power1_input") / 1e6)
            except Exception:
                return
            time.sleep(0.002)

    def __enter__(self):
        self.thread.start()
        return self

    def __exit__(self, *exc):
        self.stop = True
        self.thread.join(timeout=1.0)

    def summary(self):
        if not self.f:
            return None
        f, p = sorted(self.f[len(self.f) // 4:]), sorted(self.p[len(self.p) // 4:])      # the first quarter: the clock still settling
        return {"sclk_mhz_median": round(f[len(f) // 2]), "sclk_mhz_min_max": [round(f[0]), round(f[-1])], "power_w_median": round(p[len(p) // 2]),
                "samples": len(f)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--log-n", type=int, default=K_LOG, help="override the MSM size (parity/debug only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-create-proof", action="store_true", help="skip the configs[3] leg (create_proof simple-example k = 20)")
    ap.add_argument("--minimal", action="store_true",
                    help="only the timed commits and the NTT leg (the workload of the PMC passes: no generic / skewed / Vesta / host-pointer legs)")
    ap.add_argument("--config5", action="store_true", help="run the configs[4] leg (64 column commits + the split commit) at N = 1 too")
    ap.add_argument("--columns", type=int, default=4, help="distinct scalar columns resident in HBM")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("H2_BENCH_STREAMS", "3")),
                    help="HIP streams the independent column commits are spread over (per GPU)")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("H2_BENCH_BATCH", "1")),
                    help="column commits per h2_commit_batch_device call (the columns of a prover phase are independent, plonk/prover.rs:301-313): "
                         "K consecutive steps share one sort / accumulate / fold launch set; 1 = one h2_commit_device call per step")
    ap.add_argument("--prewarm-ms", type=float, default=400.0,
                    help="untimed commits issued before the W warm-up steps until this much wall time has passed: the chip's clocks "
                         "take a few hundred ms of sustained load to settle (the first ~20 launches of a cold run are 10-15 %% slower)")
    ap.add_argument("--lane-fraction", type=float, default=None,
                    help="share of the wave slots one accumulate launch claims (default 1)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: relaunch one rank per GPU and relay rank 0's JSON line
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", os.environ.get("MASTER_PORT", "29517"), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if rank != 0:
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)        # only rank 0 speaks on stdout (libraries print banners through C stdio)
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("H2_BENCH_BACKEND", "nccl")      # "gloo": exercise the N > 1 code path on a 1-GPU box
    if backend == "gloo":
        local_rank %= max(ndev, 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import halo2_amd as h
    from halo2_amd import fields, parallel
    from halo2_amd._lib import check
    from oracle import c_oracle as co     # input generation + CPU baseline only

    lib = h.lib()
    check(lib.h2_init(local_rank), "h2_init")
    lane_fraction = args.lane_fraction if args.lane_fraction else 1.0
    check(lib.h2_set_option(b"msm_lane_fraction", lane_fraction), "h2_set_option")
    curve = h.PALLAS
    sf = co.field_of_curve(curve, "scalar")
    n = 1 << args.log_n
    dev = torch.device("cuda", local_rank)

    # ---- synthetic inputs (seeded; identical on every rank except the per-rank columns) ----
    t0 = time.time()
    bases = co.generate_bases(curve, 0x48414C4F32, n)
    cols = [co.random_field(sf, 1000 + 97 * rank + c, n) for c in range(args.columns)]
    gen_s = time.time() - t0
    params_g = C.c_uint64(0)
    from halo2_amd.arithmetic import _p
    col_bits = int(lib.h2_commit_column_window_bits(n))       # Params::g as halo2_amd.Params registers it (17-bit windows at 2^20)
    # per-device setup, timed apart from the steps so that a 1 -> 8 GPU curve can be separated from it: the 64 MiB upload of `g`
    # from pageable host memory + the table build (15 or 16 rows of 2^(c w) multiples, 1 GiB at k = 20) -- once per Params per GPU
    torch.cuda.synchronize()
    t_reg = time.perf_counter()
    check(lib.h2_bases_register_ex(curve, _p(bases), n, h.FORM_MONTGOMERY, col_bits, C.byref(params_g)), "h2_bases_register_ex")
    torch.cuda.synchronize()
    register_ms = (time.perf_counter() - t_reg) * 1e3
    d_cols = [torch.from_numpy(c.view(np.int64)).to(dev) for c in cols]
    # the blind term of Params::commit: w = a further seeded point, r = one seeded scalar per column (commitment.rs:119-130)
    w_host = co.generate_bases(curve, 0x77, 1)[0]
    blinds_host = co.random_field(sf, 0xB11D + rank, args.columns)
    # `w` is a field of Params: installed once as the table's blind column, as halo2_amd.Params does (commits pass only r)
    check(lib.h2_bases_set_blind_base(params_g, _p(w_host), h.FORM_MONTGOMERY), "h2_bases_set_blind_base")
    d_blinds = torch.from_numpy(blinds_host.view(np.int64)).to(dev)
    d_out = torch.zeros((max(args.steps, 1), 12), dtype=torch.int64, device=dev)
    streams = [torch.cuda.current_stream()] if args.streams <= 1 else [torch.cuda.Stream(device=dev) for _ in range(args.streams)]
    sps = [C.c_void_p(s_.cuda_stream) for s_ in streams]

    def step(i):
        # consecutive column commits are independent (plonk/prover.rs:305-309): round-robin them over the streams
        c_ = i % len(d_cols)
        rc = lib.h2_commit_device(params_g, d_cols[c_].data_ptr(), n, None, d_blinds[c_].data_ptr(), h.FORM_MONTGOMERY,
                                  0, d_out[i % d_out.shape[0]].data_ptr(), sps[i % len(sps)])
        check(rc, "h2_commit_device")

    # K steps per call: the library's column-batched commit (one launch set for K independent columns, blockIdx.z = column).
    # A step is still ONE column commit with its blind; the batches go round-robin over the streams.
    K_B = max(1, min(args.batch, 8))
    vpK = C.c_void_p * K_B

    def step_batch(i0, k, j):
        cs_ = [(i0 + q) % len(d_cols) for q in range(k)]
        rc = lib.h2_commit_batch_device(params_g, (C.c_void_p * k)(*[d_cols[c_].data_ptr() for c_ in cs_]), k, n, None,
                                        (C.c_void_p * k)(*[d_blinds[c_].data_ptr() for c_ in cs_]), h.FORM_MONTGOMERY, 0,
                                        (C.c_void_p * k)(*[d_out[(i0 + q) % d_out.shape[0]].data_ptr() for q in range(k)]), sps[j % len(sps)])
        check(rc, "h2_commit_batch_device")

    def run_steps(count):
        if K_B <= 1:
            for i in range(count):
                step(i)
            return
        for j, i0 in enumerate(range(0, count, K_B)):
            step_batch(i0, min(K_B, count - i0), j)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(len(sps) * K_B)     # one untimed call per stream: each (device, stream) pair owns a workspace that is
    sync_all()                    # allocated on first use (hipMalloc synchronises the device)
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:      # clock settling, untimed (not part of W or K)
        run_steps(2 * len(sps) * K_B)
        torch.cuda.synchronize()
    sync_all()
    run_steps(args.warmup)
    sync_all()
    # HIP events around the three stages inside the timed region.  H2_BENCH_PROF_LEVEL=2 records the dominant kernel only, 0 nothing:
    # measured on one box, twice each (profiles/r04_bench_event_overhead.txt): 1066-1087 / 1071-1087 / 1087-1096 M/s -- the events cost
    # less than the run-to-run noise
    lib.h2_profile_enable(int(os.environ.get("H2_BENCH_PROF_LEVEL", "1")))
    sync_all()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    elapsed_local = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed_own = elapsed_local if world > 1 else elapsed          # this rank's own K steps, before the closing barrier
    comm_dev = dev if backend == "nccl" else torch.device("cpu")
    rank_figures = None
    if world > 1:
        # every rank's own figures travel to rank 0 (the first real 1 -> 8 curve must be readable for stragglers: a slow GPU, a slow
        # table build, a clock that one socket's power limit cut deeper): [its K steps alone, its table registration]
        mine = torch.tensor([elapsed_own, register_ms], dtype=torch.float64, device=comm_dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rank_figures = {"ms_per_step_own": [round(float(x[0]) / max(args.steps, 1) * 1e3, 4) for x in allr],
                    "bases_register_ms": [round(float(x[1]), 1) for x in allr]}
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        register_ms = max(float(x[1]) for x in allr)     # setup: the slowest rank's (8 table builds share nothing but the host's memory bandwidth)
    prof, busy = {}, {}
    for name, slot in (("msm_accumulate", 0), ("msm_sort", 2), ("msm_reduce", 3)):
        ms, bz, cnt = C.c_double(0), C.c_double(0), C.c_uint64(0)
        lib.h2_profile_read_busy(slot, C.byref(ms), C.byref(bz), C.byref(cnt))
        prof[name] = (ms.value, cnt.value)
        busy[name] = bz.value
    lib.h2_profile_enable(0)
    first = d_out[0].cpu().numpy().view(np.uint64).copy()   # output of timed step 0 = commit(cols[0], blinds[0]); d_out is reused below
    sc_full = np.ascontiguousarray(np.concatenate([cols[0], blinds_host[0:1]]))          # what Params::commit hands best_multiexp:
    bases_full = np.ascontiguousarray(np.concatenate([bases, w_host.reshape(1, 8)]))     # poly ++ [r], g ++ [w] (commitment.rs:120-129)

    # ---- the same kernel alone on the chip (one stream, full lane fraction): what the multi-stream figure dilutes ----
    iso = {}
    if rank == 0:
        check(lib.h2_set_option(b"msm_lane_fraction", 1.0), "h2_set_option")
        t_w = time.perf_counter()                  # the host work above let the clocks drop: warm up by time, untimed
        while time.perf_counter() - t_w < 0.1:
            for i in range(4):
                check(lib.h2_commit_device(params_g, d_cols[i % len(d_cols)].data_ptr(), n, None, d_blinds[i % len(d_cols)].data_ptr(), h.FORM_MONTGOMERY, 0,
                                           d_out[0].data_ptr(), sps[0]), "h2_commit_device")
            torch.cuda.synchronize()
        lib.h2_profile_enable(1)
        for i in range(20):
            c_ = i % len(d_cols)
            rc = lib.h2_commit_device(params_g, d_cols[c_].data_ptr(), n, None, d_blinds[c_].data_ptr(), h.FORM_MONTGOMERY, 0,
                                      d_out[0].data_ptr(), sps[0])
            check(rc, "h2_commit_device")
        torch.cuda.synchronize()
        for name, slot in (("msm_accumulate", 0), ("msm_sort", 2), ("msm_reduce", 3)):
            ms, cnt = C.c_double(0), C.c_uint64(0)
            lib.h2_profile_read(slot, C.byref(ms), C.byref(cnt))
            iso[name] = round(ms.value / max(cnt.value, 1), 4)
        lib.h2_profile_enable(0)
        check(lib.h2_set_option(b"msm_lane_fraction", lane_fraction), "h2_set_option")

    # ---- shader clock and socket power under the two loads above (amdgpu hwmon, sampled from a thread): the sustained rate is the
    # accumulate's issue bound AT THE CLOCK THE POWER LIMIT LEAVES, so the line carries that clock ----
    clock = None
    hw = _hwmon_dir(local_rank) if not args.minimal and (rank == 0 or world > 1) else None
    if world > 1 and not args.minimal:
        # N > 1: EVERY rank samples its own GPU for 0.3 s under the timed region's schedule, all at once (the sockets of one node draw on
        # one power envelope and one cooling loop), and rank 0 lists the medians
        own = [0.0, 0.0]
        if hw:
            sync_all()
            with _ClockSampler(hw) as smp_r:
                t_l = time.perf_counter()
                while time.perf_counter() - t_l < 0.3:
                    run_steps(max(args.steps, 1))
                    torch.cuda.synchronize()
            sm_ = smp_r.summary()
            if sm_:
                own = [float(sm_["sclk_mhz_median"]), float(sm_["power_w_median"])]
        else:
            sync_all()
        mine_c = torch.tensor(own, dtype=torch.float64, device=comm_dev)
        allc = [torch.zeros_like(mine_c) for _ in range(world)]
        dist.all_gather(allc, mine_c)
        if rank_figures is not None:
            rank_figures["sclk_mhz_median_under_timed_schedule"] = [round(float(x[0])) for x in allc]
            rank_figures["power_w_median_under_timed_schedule"] = [round(float(x[1])) for x in allc]
        hw = hw if rank == 0 else None
    if hw:
        def load(fn, seconds=0.4):
            t_l = time.perf_counter()
            cnt = 0
            with _ClockSampler(hw) as smp:
                while time.perf_counter() - t_l < seconds:
                    cnt += fn()
                    torch.cuda.synchronize()
                dt = time.perf_counter() - t_l
            r = smp.summary()
            if r:
                r["ms_per_commit"] = round(dt / max(cnt, 1) * 1e3, 4)
            return r

        def one_at_a_time():
            for i in range(20):
                c_ = i % len(d_cols)
                check(lib.h2_commit_device(params_g, d_cols[c_].data_ptr(), n, None, d_blinds[c_].data_ptr(), h.FORM_MONTGOMERY, 0, d_out[0].data_ptr(), sps[0]),
                      "h2_commit_device")
            return 20

        def timed_schedule():
            run_steps(max(args.steps, 1))
            return max(args.steps, 1)
        try:
            with open(os.path.join(hw, "power1_cap")) as fh:
                cap_w = int(fh.read().strip()) / 1e6
        except Exception:
            cap_w = None
        clock = {"source": "amdgpu hwmon freq1_input / power1_input of this GPU, sampled every ~2 ms for 0.4 s per load, after the timed region",
                 "power_cap_w": cap_w, "commits_one_at_a_time": load(one_at_a_time), "timed_region_schedule": load(timed_schedule)}

    # ---- the same multiexp WITHOUT registered bases (`best_multiexp(coeffs, bases)` as the reference calls it, bases read
    # from HBM each time, endomorphism split instead of the precomputed table): reported beside the headline ----
    generic = None
    if rank == 0 and not args.minimal:
        d_bases = torch.from_numpy(bases_full.view(np.int64)).to(dev)
        d_sc = torch.from_numpy(sc_full.view(np.int64)).to(dev)
        d_gen = torch.zeros(12, dtype=torch.int64, device=dev)
        reps_g, warm_g = 24, 4                # (round 6: 24 timed calls after 4 -- the first calls behind a synchronisation still see the clock ramp)
        for rep_ in range(reps_g + warm_g):
            if rep_ == warm_g:
                torch.cuda.synchronize()
                t6 = time.perf_counter()
            check(lib.h2_msm_device(curve, d_sc.data_ptr(), d_bases.data_ptr(), n + 1, h.FORM_MONTGOMERY, 0, d_gen.data_ptr(), sps[0]),
                  "h2_msm_device")
        torch.cuda.synchronize()
        g_ms = (time.perf_counter() - t6) / reps_g * 1e3
        generic = {"what": "h2_msm_device = best_multiexp(coeffs, bases) (arithmetic.rs:143) on n + 1 device-resident points that are NOT "
                           "registered (no precomputed table; endomorphism split), one call at a time on one stream",
                   "ms": round(g_ms, 4), "Mscalar_mults_per_s": round((n + 1) / g_ms / 1e3, 1),
                   "equals_registered_path": bool(co.jac_to_affine_ints(curve, d_gen.cpu().numpy().view(np.uint64)) == co.jac_to_affine_ints(curve, first))}
        # the same call as INDEPENDENT multiexps round-robin over the headline's streams (the shape of a prover phase without a
        # registered table): one call's sort and fold run beside another's accumulate, as for the headline's commits
        if len(sps) > 1:
            d_gens = [torch.zeros(12, dtype=torch.int64, device=dev) for _ in sps]
            reps_s = 6 * len(sps)
            for rep_ in range(reps_s + len(sps)):
                if rep_ == len(sps):
                    torch.cuda.synchronize()
                    t6 = time.perf_counter()
                j_ = rep_ % len(sps)
                check(lib.h2_msm_device(curve, d_sc.data_ptr(), d_bases.data_ptr(), n + 1, h.FORM_MONTGOMERY, 0, d_gens[j_].data_ptr(), sps[j_]),
                      "h2_msm_device")
            torch.cuda.synchronize()
            gs_ms = (time.perf_counter() - t6) / reps_s * 1e3
            generic["independent_calls"] = {"streams": len(sps), "ms_per_call": round(gs_ms, 4), "Mscalar_mults_per_s": round((n + 1) / gs_ms / 1e3, 1),
                                            "all_equal_affine": bool(all(co.jac_to_affine_ints(curve, g_.cpu().numpy().view(np.uint64)) == co.jac_to_affine_ints(curve, first)
                                                                         for g_ in d_gens))}      # (Jacobian triples differ with the order of a bucket's additions)
            del d_gens
        # ---- roofline of the generic path (review item 1): additions per call, the rate they are done at over the WHOLE call (sort, folds and the
        # Horner chain included), and that rate against the accumulate's issue bound at the clock sampled under the same load ----
        pmc_all, pmc_gen_name = _newest_pmc()
        pmc_gen = (pmc_all or {}).get(f"generic_2^{args.log_n}")
        split_carry = 0.053 if curve == h.PALLAS else 0.228      # halves whose top window carries into the ninth slice (measured)
        entries = int(2 * (n + 1) * (8 * (1 - 2.0 ** -16) + split_carry))
        rg = {"kernel": "msm_accumulate<.., 512> over the window-slice groups of one call (csrc/msm_generic.hip)", "bound": "valu",
              "entries_per_call_model": entries,
              "entries_model": "2 (n + 1) half-scalars x (8 sixteen-bit windows, zero digits excepted, + the share of halves whose top window carries into the ninth slice)",
              "issue_bound_definition": "1024 SIMDs x 64 lanes x sampled shader clock / ~7400 issue cycles per mixed addition and wave (roofline.valu.issue_bound_source)",
              "G_add_per_s_over_the_call": round(entries / (g_ms * 1e-3) / 1e9, 2),
              "G_add_per_s_independent_calls": round(entries / (generic["independent_calls"]["ms_per_call"] * 1e-3) / 1e9, 2) if "independent_calls" in generic else None,
              "algorithmic_GBps_over_the_call": round(ALGO_BYTES_PER_PAIR * (n + 1) / (g_ms * 1e-3) / 1e9, 1),
              "traffic": ((pmc_gen or {}).get("accumulate_latency_form") or {}).get("hbm_bytes_per_call"),
              "traffic_independent_calls": ((pmc_gen or {}).get("accumulate_throughput_form") or {}).get("hbm_bytes_per_call"),
              "traffic_source": f"{pmc_gen_name}: the accumulate launches of one call (FETCH_SIZE x calibration + WRITE_SIZE), not measured in this run"}
        hw_g = _hwmon_dir(local_rank)
        if hw_g:
            def _one():
                for _ in range(8):
                    check(lib.h2_msm_device(curve, d_sc.data_ptr(), d_bases.data_ptr(), n + 1, h.FORM_MONTGOMERY, 0, d_gen.data_ptr(), sps[0]), "h2_msm_device")
                return 8

            def _three():
                for i_ in range(4 * len(sps)):
                    check(lib.h2_msm_device(curve, d_sc.data_ptr(), d_bases.data_ptr(), n + 1, h.FORM_MONTGOMERY, 0, d_gen.data_ptr(), sps[i_ % len(sps)]), "h2_msm_device")
                return 4 * len(sps)
            for key_, fn_, rate_ in (("one_call_at_a_time", _one, "G_add_per_s_over_the_call"), ("independent_calls", _three, "G_add_per_s_independent_calls")):
                t_l, cnt_ = time.perf_counter(), 0
                with _ClockSampler(hw_g) as smp_g:
                    while time.perf_counter() - t_l < 0.3:
                        cnt_ += fn_()
                        torch.cuda.synchronize()
                sm_g = smp_g.summary()
                if sm_g and rg.get(rate_):
                    bound_ = 1024 * 64 * float(sm_g["sclk_mhz_median"]) * 1e6 / 7400.0 / 1e9
                    rg[key_] = {"sclk_mhz_median": sm_g["sclk_mhz_median"], "power_w_median": sm_g["power_w_median"], "issue_bound_G_add_per_s": round(bound_, 2),
                                "frac_of_issue_bound": round(rg[rate_] / bound_, 3)}
        generic["roofline_generic"] = rg
        del d_bases, d_sc

    # ---- skewed columns: 90 % zeros, and every scalar < 2^16 -- same kernels, same partition ----
    skew = {}
    if rank == 0 and not args.minimal:
        z = cols[0].copy()
        z[np.arange(n) % 10 != 0] = 0
        small = fields.to_limbs([((i * 2654435761) & 0xFFFF) for i in range(1 << 12)], sf)
        small = np.tile(small, (n >> 12, 1)) if n >= (1 << 12) else small[:n]
        for name, col in (("zeros90", z), ("below_2^16", np.ascontiguousarray(small))):
            d_c = torch.from_numpy(col.view(np.int64)).to(dev)
            for rep_ in range(3):
                if rep_ == 1:
                    torch.cuda.synchronize()
                    t4 = time.perf_counter()
                check(lib.h2_commit_device(params_g, d_c.data_ptr(), n, None, None, h.FORM_MONTGOMERY, 0, d_out[0].data_ptr(), sps[0]),
                      "h2_commit_device")
            torch.cuda.synchronize()
            skew[name] = {"ms": round((time.perf_counter() - t4) / 2 * 1e3, 4)}
            if name == "zeros90" and args.log_n <= 20:
                got = d_out[0].cpu().numpy().view(np.uint64)
                nz = np.arange(n) % 10 == 0
                want = h.best_multiexp(np.ascontiguousarray(col[nz]), np.ascontiguousarray(bases[nz]), curve)
                skew[name]["equals_dense_msm_over_nonzeros"] = bool(co.jac_to_affine_ints(curve, got) == co.jac_to_affine_ints(curve, want))
            del d_c

    # ---- parity spot check of the timed outputs (rank 0: first column vs the split-and-sum identity) ----
    parts = [] if args.minimal else [h.best_multiexp(cols[0][i * n // 4:(i + 1) * n // 4], bases[i * n // 4:(i + 1) * n // 4], curve) for i in range(4)]
    if not args.minimal:
        parts.append(h.best_multiexp(blinds_host[0:1], w_host.reshape(1, 8), curve))          # + r * w
    split_ok = None if args.minimal else co.jac_to_affine_ints(curve, h.points_sum(np.stack(parts), curve)) == co.jac_to_affine_ints(curve, first)

    # ---- multi-GPU exchange step: one range-split MSM, partials all-gathered over RCCL, summed locally ----
    split_msm_ok = None
    if world > 1:
        shared = co.random_field(sf, 4242, n)           # same column on every rank
        total = parallel.split_msm(shared, bases, curve, rank, world, device=comm_dev)
        whole = h.best_multiexp(shared, bases, curve) if rank == 0 else None
        if rank == 0:
            split_msm_ok = co.jac_to_affine_ints(curve, total) == co.jac_to_affine_ints(curve, whole)
    split_rccl_c = None       # the in-library exchange (the library's own RCCL communicator) runs LAST, see _library_rccl_legs below

    # ---- BASELINE configs[4], TIMED: 64 independent 2^20 column commits over the node (64 / world per GPU, one batched call per
    # rank, no collective) and ONE commit over the registered bases split by table-column range over the ranks, its 96-byte
    # partials exchanged by a single all-gather and summed locally.  Runs for N > 1 (and at N = 1 with --config5, where the
    # "exchange" is a gather of one).  Every figure is the max over ranks of a barrier-bracketed region.
    config5 = None
    if (world > 1 or args.config5) and not args.minimal:
        def _maxr(x):
            if world == 1:
                return float(x)
            t_ = torch.tensor([x], dtype=torch.float64, device=comm_dev)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            return float(t_.item())
        per_rank = max(1, 64 // world)
        arr = C.c_void_p * per_rank
        d_bl64 = d_blinds[[i % len(d_cols) for i in range(per_rank)]].contiguous()
        d_out64 = torch.zeros((per_rank, 12), dtype=torch.int64, device=dev)
        sc64 = arr(*[d_cols[i % len(d_cols)].data_ptr() for i in range(per_rank)])
        bl64 = arr(*[d_bl64[i].data_ptr() for i in range(per_rank)])
        o64 = arr(*[d_out64[i].data_ptr() for i in range(per_rank)])
        col_ms = []
        for rep_ in range(3):
            sync_all()
            t5 = time.perf_counter()
            check(lib.h2_commit_batch_device(params_g, sc64, per_rank, n, None, bl64, h.FORM_MONTGOMERY, 0, o64, None), "h2_commit_batch_device")
            sync_all()
            col_ms.append(_maxr((time.perf_counter() - t5) * 1e3))
        cols_ms = min(col_ms[1:])                                         # the first repetition allocates the batch streams' workspaces
        cols_ok = bool(co.jac_to_affine_ints(curve, d_out64[0].cpu().numpy().view(np.uint64)) == co.jac_to_affine_ints(curve, first))
        # per-rank PCIe: one 32 MiB column from pageable host memory (what every fresh witness column costs before it is resident)
        sync_all()
        t5 = time.perf_counter()
        d_tmp = torch.from_numpy(cols[0].view(np.int64)).to(dev)
        torch.cuda.synchronize()
        h2d_ms = _maxr((time.perf_counter() - t5) * 1e3)
        del d_tmp
        # the split commit: same column + blind on every rank
        shared_c = co.random_field(sf, 4343, n)
        d_shared = torch.from_numpy(shared_c.view(np.int64)).to(dev)
        d_shbl = torch.from_numpy(co.random_field(sf, 4344, 1).view(np.int64)).to(dev)[0].contiguous()      # the same blind on every rank
        reps5 = 12
        total5 = parallel.split_commit(params_g, d_shared, rank, world, d_shbl)      # untimed: communicator warm-up
        sync_all()
        t5 = time.perf_counter()
        for _ in range(reps5):
            total5 = parallel.split_commit(params_g, d_shared, rank, world, d_shbl)
        sync_all()
        split_ms = _maxr((time.perf_counter() - t5) / reps5 * 1e3)
        # the exchange step alone: 96 bytes per rank
        ag_us = None
        if world > 1 and backend == "nccl":
            src = torch.zeros(12, dtype=torch.int64, device=dev)
            dst = torch.zeros((world, 12), dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(dst, src)
            sync_all()
            t5 = time.perf_counter()
            for _ in range(50):
                dist.all_gather_into_tensor(dst, src)
            torch.cuda.synchronize()
            ag_us = _maxr((time.perf_counter() - t5) / 50 * 1e6)
        whole5 = torch.zeros(12, dtype=torch.int64, device=dev)
        check(lib.h2_commit_device(params_g, d_shared.data_ptr(), n, None, d_shbl.data_ptr(), h.FORM_MONTGOMERY, 0, whole5.data_ptr(), None), "h2_commit_device")
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        for _ in range(reps5):
            check(lib.h2_commit_device(params_g, d_shared.data_ptr(), n, None, d_shbl.data_ptr(), h.FORM_MONTGOMERY, 0, whole5.data_ptr(), None), "h2_commit_device")
        torch.cuda.synchronize()
        whole_ms = (time.perf_counter() - t5) / reps5 * 1e3
        split_ok5 = bool(co.jac_to_affine_ints(curve, total5.cpu().numpy().view(np.uint64)) == co.jac_to_affine_ints(curve, whole5.cpu().numpy().view(np.uint64)))
        lib_ms, lib_ok = None, None      # filled in by _library_rccl_legs at the very end
        config5 = {"what": "BASELINE configs[4]: 64 independent 2^20-point column commits (with blinds) spread over the ranks, then one commit "
                           "split by table-column range over the ranks + one 96-byte all-gather + local sum",
                   "columns_total": per_rank * world, "columns_per_gpu": per_rank, "columns_ms": round(cols_ms, 3),
                   "columns_Mscalar_mults_per_s": round(per_rank * world * n / cols_ms / 1e3, 1), "columns_first_equals_timed_step": cols_ok,
                   "h2d_ms_per_32MiB_column_max_over_ranks": round(h2d_ms, 3),
                   "split_commit_ms": round(split_ms, 4), "whole_commit_one_gpu_ms": round(whole_ms, 4), "split_equals_whole": split_ok5,
                   "allgather_96B_us": None if ag_us is None else round(ag_us, 1),
                   "split_commit_rccl_in_library_ms": None if lib_ms is None else round(lib_ms, 4), "split_commit_rccl_in_library_ok": lib_ok,
                   "exchange_backend": backend}

    # ---- NTT leg (reported beside the headline value; Fp, k = 20 and 2^22 round trip) ----
    ntt = {}
    if rank == 0:
        from oracle import pasta
        for log_n in (20, 22):
            a = co.random_field(h.FP, 7 + log_n, 1 << log_n)
            d_a = torch.from_numpy(a.view(np.int64)).to(dev)
            omega = fields.scalar_limbs(pasta.omega_for(pasta.P, log_n), h.FP)
            t_w = time.perf_counter()                # clocks settle over tens of milliseconds of sustained load (the GPU idled
            while time.perf_counter() - t_w < 0.05:  # through the CPU baseline): warm up by time, not by count
                for _ in range(25):
                    h.best_fft(d_a, omega, log_n, h.FP)
                torch.cuda.synchronize()
            reps, runs = 40, []
            for _run in range(3):
                t1 = time.perf_counter()
                for _ in range(reps):
                    h.best_fft(d_a, omega, log_n, h.FP)
                torch.cuda.synchronize()
                runs.append((time.perf_counter() - t1) / reps)
            dt = sorted(runs)[1]                     # median of three runs of 40 transforms
            lib.h2_profile_enable(1)                 # per-pass HIP events (they cost ~20 us per transform: not in `ms`)
            for _ in range(reps):
                h.best_fft(d_a, omega, log_n, h.FP)
            torch.cuda.synchronize()
            ms, cnt = C.c_double(0), C.c_uint64(0)
            lib.h2_profile_read(1, C.byref(ms), C.byref(cnt))
            lib.h2_profile_enable(0)
            bf = (1 << (log_n - 1)) * log_n
            # the reference recomputes its twiddles on every call (arithmetic.rs:215-221); the library caches the stage-major table per
            # (field, omega, log n) and every `ms` above is a cache hit: the FIRST transform with an omega the process has not seen
            # (a fresh random one, benches/fft.rs:17) pays the table build (ntt_twiddles9: 2^log_n - 1 entries of 36 B) in front of it
            fresh = co.random_field(h.FP, 0xF5E5 + log_n + int(time.time() * 1e3) % 100000, 1)[0]
            d_m = torch.from_numpy(a.view(np.int64)).to(dev)
            torch.cuda.synchronize()
            t_m = time.perf_counter()
            h.best_fft(d_m, fresh, log_n, h.FP)
            torch.cuda.synchronize()
            first_ms = (time.perf_counter() - t_m) * 1e3
            t_m = time.perf_counter()
            h.best_fft(d_m, fresh, log_n, h.FP)
            torch.cuda.synchronize()
            again_ms = (time.perf_counter() - t_m) * 1e3
            del d_m
            cpu_ntt = None
            if world == 1 and not args.no_cpu_baseline:
                cpu_runs = []
                for _ in range(5):
                    t2 = time.perf_counter()
                    ref_out = co.best_fft(h.FP, a, omega, log_n)
                    cpu_runs.append(time.perf_counter() - t2)
                cpu_dt = sorted(cpu_runs)[2]
                d_chk = torch.from_numpy(a.view(np.int64)).to(dev)
                h.best_fft(d_chk, omega, log_n, h.FP)
                torch.cuda.synchronize()
                cpu_ntt = {"ms": round(cpu_dt * 1e3, 2), "runs_ms": [round(r * 1e3, 2) for r in cpu_runs], "sample": "median of 5 runs of the same transform",
                           "Gbutterflies_per_s": round(bf / cpu_dt / 1e9, 4),
                           "kind": "port", "host_cores": int(co.lib().orc_get_threads()),
                           "bit_exact_vs_gpu": bool(np.array_equal(d_chk.cpu().numpy().view(np.uint64), ref_out))}
                del d_chk
            # independent column FFTs (prover.rs:111-117, 322-327) in one batched call (internal streams, small-tile plan)
            ms_dt = None
            if len(streams) > 1:
                d_cols_ntt = [torch.from_numpy(a.view(np.int64)).to(dev) for _ in range(2 * len(streams))]
                torch.cuda.synchronize()
                reps_b = 8
                t_w = time.perf_counter()
                while time.perf_counter() - t_w < 0.05:          # the CPU transform above left the GPU idle: warm up by time
                    h.best_fft_batch(d_cols_ntt, omega, log_n, h.FP)
                    torch.cuda.synchronize()
                t3 = time.perf_counter()
                for rep_ in range(reps_b):
                    h.best_fft_batch(d_cols_ntt, omega, log_n, h.FP)
                torch.cuda.synchronize()
                ms_dt = (time.perf_counter() - t3) / (reps_b * len(d_cols_ntt))
                del d_cols_ntt
            rt_ms = None
            if True:          # BASELINE configs[2]: forward + inverse round trip (2^22 is the config's size; 2^20 beside it)
                omega_inv = fields.scalar_limbs(pow(pasta.omega_for(pasta.P, log_n), -1, pasta.P), h.FP)
                divisor = fields.scalar_limbs(pow(1 << log_n, -1, pasta.P), h.FP)
                d_rt = torch.from_numpy(a.view(np.int64)).to(dev)
                for rep_ in range(3):
                    if rep_ == 1:
                        torch.cuda.synchronize()
                        t5 = time.perf_counter()
                    h.best_fft(d_rt, omega, log_n, h.FP)
                    check(lib.h2_ifft_device(h.FP, d_rt.data_ptr(), log_n, _p(omega_inv), _p(divisor), h.FORM_MONTGOMERY,
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)), "h2_ifft_device")
                torch.cuda.synchronize()
                rt_ms = {"ms": round((time.perf_counter() - t5) / 2 * 1e3, 4),
                         "returns_input": bool(np.array_equal(d_rt.cpu().numpy().view(np.uint64), a))}
                del d_rt
            ntt[f"2^{log_n}"] = {"ms": round(dt * 1e3, 4), "forward_inverse_roundtrip": rt_ms,
                                 "twiddle_miss": {"first_call_fresh_omega_ms": round(first_ms, 4), "second_call_same_omega_ms": round(again_ms, 4),
                                                  "twiddle_miss_ms": round(first_ms - again_ms, 4),
                                                  "what": "one synchronised best_fft with an omega this process has not used (the table build, ntt_twiddles9, runs in "
                                                          "front of the passes), then the same call again (a cache hit; a lone synchronised call, so above `ms`)"}, "Gbutterflies_per_s": round(bf / dt / 1e9, 3), "cpu_baseline": cpu_ntt,
                                 "kernel_ms": round(ms.value / reps, 4), "passes": int(cnt.value // reps),
                                 "independent_columns": None if ms_dt is None else {
                                     "streams": len(streams), "ms_per_fft": round(ms_dt * 1e3, 4), "Gbutterflies_per_s": round(bf / ms_dt / 1e9, 3)},
                                 "algorithmic_GBps": round(64.0 * (1 << log_n) / dt / 1e9, 1)}
            del d_a

    # ---- CPU baseline: C restatement of the reference algorithm on this box's host cores ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cores = co.lib().orc_get_threads()
        co.best_multiexp(curve, sc_full, bases_full)                      # warm-up run (page faults, thread pool)
        runs = []
        for _ in range(10):
            t1 = time.perf_counter()
            ref = co.best_multiexp(curve, sc_full, bases_full)
            runs.append(time.perf_counter() - t1)
        cpu_s = sorted(runs)[len(runs) // 2]
        cpu_ok = co.jac_to_affine_ints(curve, ref) == co.jac_to_affine_ints(curve, first)
        c = co.lib().orc_window_bits(n + 1)
        cpu = {"value": round((n + 1) / cpu_s / 1e6, 4), "unit": "Mscalar-mults/s", "cores": int(min(cores, 256 // c + 1)),
               "host_cores": int(cores), "kind": "port",
               "sample": f"median of 10 runs (after 1 warm-up) of the same commit as best_multiexp over 2^{args.log_n} + 1 Pallas points "
                         f"(c={c}, {256 // c + 1} window tasks, one thread each), {cpu_s:.3f} s median, {min(runs):.3f}-{max(runs):.3f} s range; "
                         "C restatement of arithmetic.rs:143-180, not the Rust reference",
               "runs_s": [round(r, 4) for r in runs], "bit_exact_vs_gpu": bool(cpu_ok)}

    # ---- first-class companions of the headline (the bench line must carry them) ----
    extra = {}
    if rank == 0 and not args.minimal:
        # (1) the Vesta commit: the curve every reference proof commits on (benches/plonk.rs:6); same kernels, other moduli
        vs = co.field_of_curve(h.VESTA, "scalar")
        v_bases = co.generate_bases(h.VESTA, 0x56455354, n)
        v_cols = [co.random_field(vs, 2000 + c_, n) for c_ in range(2)]
        v_w = co.generate_bases(h.VESTA, 0x78, 1)[0]
        v_bl = co.random_field(vs, 0xB11E, 2)
        hv = C.c_uint64(0)
        check(lib.h2_bases_register_ex(h.VESTA, _p(v_bases), n, h.FORM_MONTGOMERY, col_bits, C.byref(hv)), "h2_bases_register_ex")
        dv_cols = [torch.from_numpy(c_.view(np.int64)).to(dev) for c_ in v_cols]
        check(lib.h2_bases_set_blind_base(hv, _p(v_w), h.FORM_MONTGOMERY), "h2_bases_set_blind_base")
        dv_bl = torch.from_numpy(v_bl.view(np.int64)).to(dev)
        reps_v, rep_ = 60, 0

        def v_commit(r_):
            check(lib.h2_commit_device(hv, dv_cols[r_ % 2].data_ptr(), n, None, dv_bl[r_ % 2].data_ptr(), h.FORM_MONTGOMERY, 0,
                                       d_out[r_ % d_out.shape[0]].data_ptr(), sps[r_ % len(sps)]), "h2_commit_device")
        t_w = time.perf_counter()                  # the GPU idled through the CPU baseline: warm up by time, as the headline does
        while time.perf_counter() - t_w < max(args.prewarm_ms, 50) * 1e-3:
            for _ in range(2 * len(sps)):
                v_commit(rep_)
                rep_ += 1
            torch.cuda.synchronize()
        t7 = time.perf_counter()
        for _ in range(reps_v):
            v_commit(rep_)
            rep_ += 1
        torch.cuda.synchronize()
        v_ms = (time.perf_counter() - t7) / reps_v * 1e3
        v_last = d_out[(rep_ - 1) % d_out.shape[0]].cpu().numpy().view(np.uint64).copy()
        k_ = (rep_ - 1) % 2
        v_want = h.best_multiexp(np.ascontiguousarray(np.concatenate([v_cols[k_], v_bl[k_:k_ + 1]])),
                                 np.ascontiguousarray(np.concatenate([v_bases, v_w.reshape(1, 8)])), h.VESTA)
        extra["vesta_commit"] = {"ms_per_commit": round(v_ms, 4), "Mscalar_mults_per_s": round(n / v_ms / 1e3, 1), "streams": len(sps),
                                 "equals_generic_multiexp": bool(co.jac_to_affine_ints(h.VESTA, v_last) == co.jac_to_affine_ints(h.VESTA, v_want))}
        lib.h2_bases_free(hv)
        del dv_cols, v_bases
        # (2) end to end from HOST memory: h2_commit copies the 32 MiB column over PCIe, commits, copies the point back
        out_h = np.zeros(12, dtype=np.uint64)
        e2e = []
        for rep_ in range(6):
            t8 = time.perf_counter()
            check(lib.h2_commit(params_g, _p(cols[rep_ % len(cols)]), n, _p(w_host), _p(blinds_host[rep_ % len(cols)]), h.FORM_MONTGOMERY, 0,
                                _p(out_h)), "h2_commit")
            e2e.append(time.perf_counter() - t8)
        e2e_ms = sorted(e2e[1:])[len(e2e[1:]) // 2] * 1e3
        extra["host_pointer_commit"] = {"what": "h2_commit: pageable host scalars -> PCIe in 8 MiB ranges, each committed as it lands -> one fold -> 96 B back; one call at a time (SURVEY 8d 'end-to-end including H2D')",
                                        "ms": round(e2e_ms, 4), "Mscalar_mults_per_s": round(n / e2e_ms / 1e3, 1)}
        # (2b) the literal seam of INTEGRATION.md section 2 at the headline size: best_multiexp -> h2_msm and best_fft -> h2_ntt with HOST
        # pointers, both PCIe directions inside the call (pageable memory, one call at a time, median of 5 after one warm-up call)
        def _med(f, reps=5):
            f()
            ts = []
            for _ in range(reps):
                t_ = time.perf_counter()
                f()
                ts.append(time.perf_counter() - t_)
            return sorted(ts)[len(ts) // 2] * 1e3
        msm_h = _med(lambda: check(lib.h2_msm(curve, _p(cols[0]), _p(bases), n, h.FORM_MONTGOMERY, 0, _p(out_h)), "h2_msm"))
        from oracle import pasta as _pasta
        host_ntt = {}
        for log_n in (20, 22):
            a_h = co.random_field(h.FP, 70 + log_n, 1 << log_n)
            om = fields.scalar_limbs(_pasta.omega_for(_pasta.P, log_n), h.FP)
            host_ntt[f"2^{log_n}"] = round(_med(lambda: check(lib.h2_ntt(h.FP, _p(a_h), log_n, _p(om), h.FORM_MONTGOMERY), "h2_ntt")), 4)
        extra["host_pointer_seam"] = {"what": "the free functions as the Rust shim calls them: host slices in, host results out (SURVEY 8b); thresholds and the "
                                              "crossover table: INTEGRATION.md section 2, profiles/r03_crossover.json",
                                      "h2_msm_2^20_ms": round(msm_h, 4), "h2_msm_2^20_Mscalar_mults_per_s": round(n / msm_h / 1e3, 1),
                                      "h2_ntt_ms": host_ntt, "h2_commit_2^20_ms": round(e2e_ms, 4)}
        # (3) BASELINE configs[3]: create_proof of the reference's simple-example circuit at k = 20 (examples/simple_example.py:
        # product prover + product verifier; columns, quotient FFTs, multi-point opening and the opening argument all on this GPU)
        if args.log_n == 20 and not args.no_create_proof:
            import importlib.util
            spec = importlib.util.spec_from_file_location("simple_example", os.path.join(ROOT, "examples", "simple_example.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            pv = co.generate_bases(h.VESTA, 0x56455354, n + 2)          # g (n points), w, u: seeded, as everywhere in this file
            t9 = time.perf_counter()
            prm = h.Params.from_generators(h.VESTA, args.log_n, np.ascontiguousarray(pv[:n]), None, pv[n], pv[n + 1])
            torch.cuda.synchronize()
            params_s = time.perf_counter() - t9
            res = mod.prove_and_verify(prm, quiet=True)
            # (3b) the opening argument alone (poly/commitment/prover.rs:26-151) through its one-call entry points, the randomness drawn
            # beforehand (the rng is the caller's): p_poly resident (h2_open_device; the n random coefficients of s_poly still cross PCIe
            # inside, as they come from a host rng) and p_poly + s_poly as host vectors (h2_open: what a Rust caller holds)
            from halo2_amd.opening import create_proof as _open
            from halo2_amd.transcript import Blake2bWrite as _Tr
            o_px = co.random_field(vs, 0x09E1, n)
            o_pool = co.random_field(vs, 0x09E2, n + 1 + 2 * args.log_n)
            o_blind, o_x = h.Blind(co.random_field(vs, 0x09E3, 1)[0]), co.random_field(vs, 0x09E4, 1)[0]
            d_opx = torch.from_numpy(o_px.view(np.int64)).to(dev)

            def _opening(p_):
                pos = [0]

                def rng_(count):
                    pos[0] += count
                    return o_pool[pos[0] - count: pos[0]]
                tr_ = _Tr(h.VESTA)
                _open(prm, rng_, tr_, p_, o_blind, o_x)
                torch.cuda.synchronize()
                return tr_.finalize()
            o_bytes = [None, None]

            def _timed_opening(slot, p_):
                o_bytes[slot] = _opening(p_)
            open_res_ms = _med(lambda: _timed_opening(0, d_opx), reps=3)
            open_host_ms = _med(lambda: _timed_opening(1, o_px), reps=3)
            extra["opening_argument_k20"] = {
                "what": "commitment::create_proof (poly/commitment/prover.rs:26-151) for one 2^20-coefficient polynomial on Vesta as ONE library call, median of 3 "
                        "after a warm-up: the S commitment, 20 rounds (6 over the registered generators, the read-out and registration of G'_6, 14 over its table as paired commits with 8-bit sub-digits), "
                        "40 points and 2 scalars to a BLAKE2b transcript",
                "resident_p_poly_ms": round(open_res_ms, 3), "host_vectors_ms": round(open_host_ms, 3), "same_proof_bytes": bool(o_bytes[0] == o_bytes[1]),
                "proof_bytes": len(o_bytes[0])}
            del d_opx
            prm.close()
            extra["create_proof_simple_example_k20"] = {
                "accepted_and_wrong_instance_rejected": res["ok"], "create_proof_s": round(res["create_proof_s"], 4),
                "create_proof_from_host_columns_s": round(res["create_proof_from_host_columns_s"], 4),
                "host_advice_columns": res["advice_columns"],
                "create_proof_first_call_s": round(res["create_proof_first_s"], 4), "keygen_s": round(res["keygen_s"], 4),
                "verify_proof_s": round(res["verify_proof_s"], 4), "proof_bytes": res["proof_bytes"],
                "params_from_generators_s": round(params_s, 3),
                "what": "examples/simple_example.py: the reference's simple-example circuit (examples/simple-example.rs) at k = 20 on Vesta, "
                        "columns already assigned (create_proof_s: resident in HBM; create_proof_from_host_columns_s: the advice columns start in pageable host memory, "
                        "as after the reference's CPU synthesis, plonk/prover.rs:284-313, and their upload is inside the timed region); instance / advice commits, permutation, vanishing argument (quotient FFTs at 2^21), "
                        "multi-point opening and opening argument on this GPU; create_proof_s = second proof of the process"}

    if rank == 0:
        # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
        # collected separately, corrected as MI355X_MICROARCH.md prescribes); null when the workload differs
        traffic = None
        try:
            pmc, pmc_name = _newest_pmc()                                                                    # newest round wins
            if args.log_n == 20:
                t_ = pmc["msm_accumulate_2^20"]
                traffic = int(t_.get("fetch_bytes_corrected", t_["fetch_bytes_reported_max"]) + t_["write_bytes_max"])
        except Exception:
            pmc, pmc_name = None, None
        total_mults = float(n) * args.steps * world
        value = total_mults / elapsed / 1e6
        acc_ms, acc_cnt = prof["msm_accumulate"]
        # device time per launch = union of the launch intervals / launches: launches from different streams overlap, and summing
        # their durations would count shared time once per launch (r1: 1.82 ms "per launch" inside a 1.33 ms step)
        avg_ms = busy["msm_accumulate"] / max(acc_cnt, 1)
        sum_ms = acc_ms / max(acc_cnt, 1)
        cols_per_launch = args.steps / max(acc_cnt, 1)          # a batched commit accumulates its K columns in ONE launch (blockIdx.z = column)
        achieved = ALGO_BYTES_PER_PAIR * (n + 1) * cols_per_launch / (avg_ms * 1e-3) / 1e9 if acc_cnt else None
        madds = (255 // col_bits + (1 if 255 % col_bits else 0)) * (n + 1)      # per COLUMN; non-zero digits per scalar: 16 at 16 bits, 15 at 17 (255 = 15 x 17)
        out = {
            "metric": "Pallas MSM Mscalar-mults/s (+ Fp NTT Gbutterflies/s) at k=20",
            "value": round(value, 3), "unit": "Mscalar-mults/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u256 (8x32-bit Montgomery limbs)",
            "data": "synthetic",
            "config": {"workload": f"2^{args.log_n}-point Pallas best_multiexp, uniform random Fq scalars, "
                                   "bases resident (Params::g registered), one column commit WITH its blind term per step per GPU",
                       "window_bits": col_bits, "columns_resident": args.columns, "streams": len(streams),
                       "columns_per_call": K_B, "columns_per_call_note": "K consecutive steps = K independent columns handed to h2_commit_batch_device in one call "
                                                                         "(plonk/prover.rs:301-313 commits a phase's columns together); every step is still one full 2^20 commit with its blind",
                       "msm_lane_fraction": lane_fraction,
                       "parallelism": f"{world} GPU(s) x {len(streams)} stream(s) of independent column commits"},
            "roofline": {"bound": "hbm", "kernel": "msm_accumulate", "achieved": round(achieved, 2) if achieved else None,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5) if achieved else None,
                         "traffic": int(traffic * cols_per_launch) if traffic else traffic,
                         "traffic_source": f"{pmc_name}: the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel on this workload (bench/tools/r06_pmc.sh on the "
                                           "benchmarked tree), corrected by the calibration measured in the same session -- a constant read from that file, NOT measured in this run",
                         "avg_kernel_ms": round(avg_ms, 4), "launches": int(acc_cnt),
                         "columns_per_launch": round(cols_per_launch, 3),
                         "avg_kernel_ms_definition": "union of the launch intervals on the device / launches (HIP events on the launching "
                                                     "streams, timed region only); overlapped_launch_ms = plain mean of the launch durations",
                         "overlapped_launch_ms": round(sum_ms, 4), "kernel_ms_isolated": iso.get("msm_accumulate"),
                         "valu": {"madd_per_launch": int(madds * cols_per_launch), "madd_per_launch_definition": "columns_per_launch x non-zero digits of the column: 16 per scalar at 16-bit windows, 15 at 17 bits (255 = 15 x 17; the "
                                  "top window of a scalar below q < 2^254 + 2^126 never exceeds 2^16, so the recode carries nothing out of it)",
                                  "achieved_Gmadd_per_s": round(madds * cols_per_launch / (avg_ms * 1e-3) / 1e9, 2) if acc_cnt else None,
                                  "isolated_Gmadd_per_s": round(madds / (iso["msm_accumulate"] * 1e-3) / 1e9, 2) if iso.get("msm_accumulate") else None,
                                  "modmul_per_madd": 10, "montgomery_reductions_per_madd": 9, "v_mad_i64_i32_per_madd": 1151, "instructions_per_madd": 1710,
                                  "issue_bound_Gmadd_per_s": {"at_2.31_GHz_kernel_alone": 20.4, "at_1.92_GHz_sustained": 17.0},
                                  "issue_bound_source": "the mixed addition is ~1710 instructions on its common path (hipcc -S), 1151 of them v_mad_i64_i32 (8 products, 2 squares, "
                                                        "Y3's two products under ONE reduction); a SIMD issues one v_mad_i64_i32 per 4.85 cycles and the other instructions at 2.7-4.9 "
                                                        "(bench/ubench_valu.hip, profiles/r04_ubench_valu.txt): ~7400 cycles per addition and wave, 1024 SIMDs x 64 lanes x clock / 7400. "
                                                        "The clock is what the socket power limit leaves (rocm-smi under load, profiles/r04_clock_power.txt): 2.31 GHz while commits run "
                                                        "one at a time, 1.92 GHz at ~1390 W once accumulates of several streams keep the SIMDs busy without a gap -- the sustained rate of "
                                                        "the timed region is the issue bound at that clock (DESIGN.md section 4.3)"},
                         "note": "VALU integer-multiply bound, not HBM bound (DESIGN.md section 4.3): the HBM fraction is reported as the contract "
                                 "asks, the VALU figures are what track kernel quality; traffic = PMC bytes of the registered-bases path (FETCH_SIZE calibrated on 64-byte random gathers, "
                                 "the PMC file named in traffic_source), which gathers 15 precomputed multiples per point from a 1 GiB table by design: a 64-byte point "
                                 "is half a 128-byte line, and the line is what moves"},
            # the HBM-bound part of the path (north star: "bucket-scan kernel"): the two-pass bucket sort streams 244 B per scalar
            # (32 B read twice by pass 1; 15 entries x 4 B written by pass 1, read once and written once by the one-launch pass 2);
            # time = the sort stage alone on one stream
            "roofline_bucket_sort": (lambda ms_, tr_: {
                "bound": "hbm", "kernels": "msm_s1_count / _prefix / _scatter, msm_s2_bins",
                "algorithmic_bytes_per_scalar": 244,
                "achieved": round(244.0 * n / (ms_ * 1e-3) / 1e9, 1) if ms_ else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(244.0 * n / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms_ else None,
                "traffic": tr_, "traffic_source": f"{pmc_name}, not measured in this run", "stage_ms_isolated": ms_})(
                iso.get("msm_sort"), (pmc or {}).get("msm_bucket_sort_2^20", {}).get("total_hbm_bytes_corrected") if args.log_n == 20 else None),
            # the second hot kernel of the path (BASELINE configs[2]): the NTT passes, priced the same way -- 64 B per element per
            # transform over the transform's wall time, against HBM; traffic = PMC bytes of the two passes
            # (FETCH_SIZE calibrated on the passes' own access patterns, in the PMC file's own session)
            "roofline_ntt": (lambda e_: None if not e_ else {
                "bound": "hbm", "kernel": "ntt_pass9 (two passes of 10 stages at 2^20)", "achieved": e_["algorithmic_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(e_["algorithmic_GBps"] / HBM_PEAK_GBS, 5), "ms_per_transform": e_["ms"], "kernel_ms_per_transform": e_["kernel_ms"],
                "traffic": (pmc or {}).get("ntt_2^20", {}).get("total_hbm_bytes_corrected"),
                "traffic_source": f"{pmc_name}, not measured in this run",
                "note": "VALU-bound like the MSM: 10.5 M modular multiplications per 2^20 transform at ~200 G/s are 0.052 ms before any addition, carry pass "
                        "or LDS round trip; the passes issue ~4800 instructions per lane and pass (DESIGN.md section 5.5)"})(ntt.get("2^20")),
            "kernel_ms_per_step": {k: round(v[0] / max(v[1], 1), 4) for k, v in prof.items() if v[1]},
            "kernel_ms_isolated": iso,
            "clock": clock, "generic_best_multiexp": generic, "extra": extra, "skewed_columns": skew, "ntt": ntt, "cpu_baseline": cpu,
            "checks": {"split_sum_identity": None if split_ok is None else bool(split_ok), "split_msm_allgather": split_msm_ok, "split_msm_rccl_in_library": split_rccl_c},
            "config5": config5,
            "per_rank": rank_figures,
            "setup": {"bases_register_ms_max_over_ranks": round(register_ms, 1),
                      "what": "h2_bases_register_ex per GPU, untimed by the steps: 64 MiB of `g` from pageable host memory + the precomputed table "
                              f"({255 // col_bits + (1 if 255 % col_bits else 0)} rows of {n} + 1 points, 64 B each); once per Params per device"},
            "input_gen_s": round(gen_s, 2),
        }

    # ---- the exchange step INSIDE the library (its own RCCL communicator: h2_rccl_init, one 96-byte ncclAllGather, local sum) ----
    # Verification + one timing, and the only leg of this file that has never met more than one rank on hardware -- so it runs LAST,
    # when every other figure of the line is already computed, in a watchdog thread, and NOTHING collective follows it: a communicator
    # that fails to form, or forms on some ranks only, can cost this leg its two fields but not the line.  (Where it used to sit --
    # before the config-5 timings -- a late rank would have left its thread issuing torch.distributed calls beside the main thread's.)
    rccl_hung = rccl_leg_ran = False
    if world > 1 and backend == "nccl" and os.environ.get("H2_BENCH_RCCL_C", "1") != "0" and config5 is not None:
        import threading
        box = {}
        rccl_leg_ran = True
        dist.barrier()           # rank 0 arrives from its solo legs (NTT, Vesta, host seam, create_proof): enter the leg together
        torch.cuda.synchronize()

        def _library_rccl_legs():
            try:
                torch.cuda.set_device(local_rank)          # the current device is per host thread
                parallel.rccl_init(rank, world)
                d_sh = torch.from_numpy(shared.view(np.int64)).to(dev)
                d_bs = torch.from_numpy(bases.view(np.int64)).to(dev)
                out_c = parallel.split_msm_rccl(d_sh, d_bs, curve)
                torch.cuda.synchronize()
                box["msm_ok"] = bool(co.jac_to_affine_ints(curve, out_c.cpu().numpy().view(np.uint64)) == co.jac_to_affine_ints(curve, total))
                o_ = parallel.split_commit_rccl(params_g, d_shared, d_shbl)
                torch.cuda.synchronize()
                t_ = time.perf_counter()
                for _ in range(reps5):
                    o_ = parallel.split_commit_rccl(params_g, d_shared, d_shbl)
                torch.cuda.synchronize()
                box["commit_ms"] = (time.perf_counter() - t_) / reps5 * 1e3
                box["commit_ok"] = bool(co.jac_to_affine_ints(curve, o_.cpu().numpy().view(np.uint64)) == co.jac_to_affine_ints(curve, whole5.cpu().numpy().view(np.uint64)))
                parallel.rccl_finalize()
                box["done"] = True
            except Exception as exc:           # reported, never fatal
                box["error"] = f"error: {exc}"
                box["done"] = True
        th = threading.Thread(target=_library_rccl_legs, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("H2_BENCH_RCCL_C_TIMEOUT", "90")))
        rccl_hung = not box.get("done", False)
        if rank == 0:
            miss = "timeout" if rccl_hung else box.get("error")
            out["checks"]["split_msm_rccl_in_library"] = box.get("msm_ok", miss)
            out["config5"]["split_commit_rccl_in_library_ok"] = box.get("commit_ok", miss)
            if "commit_ms" in box:
                out["config5"]["split_commit_rccl_in_library_ms"] = round(box["commit_ms"], 4)
    if rank == 0:
        final_line = json.dumps(out)
    if rank == 0:
        # The JSON line is the LAST thing on stdout: RCCL (torch's communicator, the library's own) prints a version banner through
        # C stdio, which sits in the C buffer until the process exits -- i.e. it would land AFTER a line printed from Python.  Flush
        # the C buffers first, print, then point fd 1 at /dev/null so that nothing a library says at teardown can follow the line.
        import ctypes
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(final_line)
        sys.stdout.flush()
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    if not rccl_hung:
        lib.h2_bases_free(params_g)
        if world > 1:
            if not rccl_leg_ran:
                dist.barrier()                    # (after the in-library leg nothing collective follows: a peer may have left through the branch below)
            dist.destroy_process_group()
    if rccl_hung:
        os._exit(0)          # a thread is stuck inside a collective: no interpreter / communicator teardown, the line is out


if __name__ == "__main__":
    main()
