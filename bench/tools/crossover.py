#!/usr/bin/env python3
"""The literal drop-in seam, measured: `best_multiexp` -> h2_msm and `best_fft` -> h2_ntt with HOST pointers (INTEGRATION.md
section 2), over the sizes the reference's own benches sweep (benches/msm.rs k = 8..15 -> here 3..20, benches/fft.rs k = 3..18
-> here 3..22), beside the C restatement of the reference's CPU algorithm on this box's host cores.  The crossover sizes are
what `gpu::MSM_THRESHOLD` / `gpu::NTT_THRESHOLD_LOG` of the Rust shim should be set to.

Every GPU figure includes both PCIe directions (pageable host memory, one call at a time, median of several calls)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def med(f, reps):
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t)
    return sorted(ts)[len(ts) // 2] * 1e3


def main():
    import halo2_amd as h
    from halo2_amd import fields
    from halo2_amd.arithmetic import _p
    from oracle import c_oracle as co
    from oracle import pasta
    lib = h.lib()
    assert lib.h2_init(0) == 0
    curve = h.VESTA                                  # the curve every reference proof commits on
    sf = co.field_of_curve(curve, "scalar")
    kmax = 20
    bases = co.generate_bases(curve, 0xC0, 1 << kmax)
    scal = co.random_field(sf, 0xC1, 1 << kmax)
    out = np.zeros(12, dtype=np.uint64)
    all_threads = int(co.lib().orc_get_threads())
    res = {"host_threads": all_threads, "msm": {}, "fft": {},
           "threshold_rule": "smallest size from which the GPU call (both PCIe directions included) beats the FASTER of the two CPU figures "
                             "(all threads / one thread) at every larger measured size"}

    def best_cpu(e):
        return min(v for v in (e["cpu_ms"], e["cpu_1thread_ms"]) if v is not None)
    h.best_multiexp(scal[:1024], bases[:1024], curve)      # context creation
    for k in list(range(3, 21)):
        n = 1 << k
        s, b = np.ascontiguousarray(scal[:n]), np.ascontiguousarray(bases[:n])
        reps = 9 if k <= 16 else 5
        gpu = med(lambda: lib.h2_msm(curve, _p(s), _p(b), n, h.FORM_MONTGOMERY, 0, _p(out)), reps)
        cpu = med(lambda: co.best_multiexp(curve, s, b), 5 if k <= 16 else 3)
        cpu1 = None
        if k <= 14:        # one thread: the serial branch, no thread start-up (the restatement spawns its threads per call, rayon keeps a pool)
            co.lib().orc_set_threads(1)
            cpu1 = med(lambda: co.best_multiexp(curve, s, b), 3)
            co.lib().orc_set_threads(all_threads)
        res["msm"][k] = {"gpu_host_ptr_ms": round(gpu, 4), "cpu_ms": round(cpu, 4), "cpu_1thread_ms": None if cpu1 is None else round(cpu1, 4)}
    # the first size from which the GPU wins at every larger measured size
    thr = None
    for k in sorted(res["msm"], reverse=True):
        if res["msm"][k]["gpu_host_ptr_ms"] < best_cpu(res["msm"][k]):
            thr = k
        else:
            break
    res["MSM_THRESHOLD_log2"] = thr
    for k in list(range(3, 23)):
        n = 1 << k
        a = co.random_field(h.FP, 0xF0 + k, n)
        omega = fields.scalar_limbs(pasta.omega_for(pasta.P, k), h.FP)
        buf = a.copy()
        reps = 9 if k <= 18 else 5
        gpu = med(lambda: lib.h2_ntt(h.FP, _p(buf), k, _p(omega), h.FORM_MONTGOMERY), reps)
        cpu = med(lambda: co.best_fft(h.FP, a, omega, k), 5 if k <= 18 else 3)
        cpu1 = None
        if k <= 16:
            co.lib().orc_set_threads(1)
            cpu1 = med(lambda: co.best_fft(h.FP, a, omega, k), 3)
            co.lib().orc_set_threads(all_threads)
        res["fft"][k] = {"gpu_host_ptr_ms": round(gpu, 4), "cpu_ms": round(cpu, 4), "cpu_1thread_ms": None if cpu1 is None else round(cpu1, 4)}
    thr = None
    for k in sorted(res["fft"], reverse=True):
        if res["fft"][k]["gpu_host_ptr_ms"] < best_cpu(res["fft"][k]):
            thr = k
        else:
            break
    res["NTT_THRESHOLD_LOG"] = thr
    # registered commit from a host column (Params::commit as the reference calls it), 2^16 .. 2^20
    res["commit"] = {}
    for k in (12, 14, 16, 18, 20):
        n = 1 << k
        hd = __import__("ctypes").c_uint64(0)
        b = np.ascontiguousarray(bases[:n])
        s = np.ascontiguousarray(scal[:n])
        w = np.ascontiguousarray(bases[(1 << kmax) - 1])
        bl = np.ascontiguousarray(scal[5])
        wb = int(lib.h2_commit_column_window_bits(n))
        assert lib.h2_bases_register_ex(curve, _p(b), n, h.FORM_MONTGOMERY, wb, __import__("ctypes").byref(hd)) == 0
        assert lib.h2_bases_set_blind_base(hd, _p(w), h.FORM_MONTGOMERY) == 0
        lib.h2_commit(hd, _p(s), n, None, _p(bl), h.FORM_MONTGOMERY, 0, _p(out))
        gpu = med(lambda: lib.h2_commit(hd, _p(s), n, None, _p(bl), h.FORM_MONTGOMERY, 0, _p(out)), 9)
        cpu = med(lambda: co.commit(curve, b, w, s, bl), 3)
        res["commit"][k] = {"gpu_host_ptr_ms": round(gpu, 4), "cpu_ms": round(cpu, 4)}
        lib.h2_bases_free(hd)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
