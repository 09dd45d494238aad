#!/usr/bin/env python3
"""Window width of column tables (H2_COLUMN_C) against commit throughput and the stage times of one lone commit: run once per
width, e.g.  for c in 17 18 19 20; do H2_COLUMN_C=$c python bench/tools/column_c_sweep.py; done"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import halo2_amd as h
    from halo2_amd.arithmetic import _p
    from oracle import c_oracle as co
    lib = h.lib()
    assert lib.h2_init(0) == 0
    curve, n = h.PALLAS, 1 << 20
    sf = co.field_of_curve(curve, "scalar")
    bases = co.generate_bases(curve, 0x48414C4F32, n)
    cols = [co.random_field(sf, 1000 + c, n) for c in range(4)]
    w = np.ascontiguousarray(co.generate_bases(curve, 0x77, 1)[0])
    blinds = co.random_field(sf, 0xB11D, 4)
    hd = C.c_uint64(0)
    cb = int(lib.h2_commit_column_window_bits(n))
    t0 = time.perf_counter()
    assert lib.h2_bases_register_ex(curve, _p(bases), n, h.FORM_MONTGOMERY, cb, C.byref(hd)) == 0
    reg_s = time.perf_counter() - t0
    assert lib.h2_bases_set_blind_base(hd, _p(w), h.FORM_MONTGOMERY) == 0
    dev = torch.device("cuda", 0)
    d_cols = [torch.from_numpy(c.view(np.int64)).to(dev) for c in cols]
    d_bl = torch.from_numpy(blinds.view(np.int64)).to(dev)
    d_out = torch.zeros((8, 12), dtype=torch.int64, device=dev)
    nst = int(os.environ.get("SWEEP_STREAMS", "3"))
    use_blind = os.environ.get("SWEEP_BLIND", "1") != "0"
    streams = [torch.cuda.Stream(device=dev) for _ in range(nst)]
    sps = [C.c_void_p(s.cuda_stream) for s in streams]

    def step(i, sp):
        rc = lib.h2_commit_device(hd, d_cols[i % 4].data_ptr(), n, None, d_bl[i % 4].data_ptr() if use_blind else None, h.FORM_MONTGOMERY, 0,
                                  d_out[i % 8].data_ptr(), sp)
        assert rc == 0, lib.h2_last_error()
    for i in range(6):
        step(i, sps[i % nst])
    torch.cuda.synchronize()
    got = co.jac_to_affine_ints(curve, d_out[0].cpu().numpy().view(np.uint64))
    want = co.jac_to_affine_ints(curve, co.commit(curve, bases, w, cols[0], blinds[0]) if use_blind else co.best_multiexp(curve, cols[0], bases))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        for i in range(6):
            step(i, sps[i % nst])
        torch.cuda.synchronize()
    res = {"window_bits": cb, "register_s": round(reg_s, 3), "bit_exact": got == want}
    for steps in (20, 100):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i, sps[i % nst])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        res[f"ms_per_commit_{steps}steps_3streams"] = round(ms, 4)
        res[f"Mscalar_mults_per_s_{steps}"] = round(n / ms / 1e3, 1)
    lib.h2_profile_enable(1)
    for i in range(20):
        step(i, sps[0])
    torch.cuda.synchronize()
    for name, slot in (("accumulate", 0), ("sort", 2), ("fold", 3)):
        ms, cnt = C.c_double(0), C.c_uint64(0)
        lib.h2_profile_read(slot, C.byref(ms), C.byref(cnt))
        res[f"{name}_ms_isolated"] = round(ms.value / max(cnt.value, 1), 4)
    lib.h2_profile_enable(0)
    t0 = time.perf_counter()
    for i in range(20):
        step(i, sps[0])
    torch.cuda.synchronize()
    res["lone_commit_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
    # a degenerate column: every scalar equal (15 buckets take everything): the one-workgroup-per-bin pass 2 scatters such bins directly
    from halo2_amd import fields
    eq = np.ascontiguousarray(np.tile(fields.scalar_limbs(0xDEADBEEFCAFE0123456789, sf), (n, 1)))
    d_eq = torch.from_numpy(eq.view(np.int64)).to(dev)
    for rep in range(4):
        if rep == 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        assert lib.h2_commit_device(hd, d_eq.data_ptr(), n, None, None, h.FORM_MONTGOMERY, 0, d_out[0].data_ptr(), sps[0]) == 0
    torch.cuda.synchronize()
    res["all_equal_column_ms"] = round((time.perf_counter() - t0) / 3 * 1e3, 4)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
