#!/usr/bin/env python3
"""h2_commit from a host column at 2^20: range size sweep of the pipelined transfer, beside the raw PCIe figures of this box
(pageable / registered-in-place / pinned copies of 4 and 32 MiB)."""
import ctypes as C
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
    return round(sorted(ts)[len(ts) // 2] * 1e3, 4)


def main():
    import torch
    import halo2_amd as h
    from halo2_amd.arithmetic import _p
    from oracle import c_oracle as co
    lib = h.lib()
    assert lib.h2_init(0) == 0
    hip = C.CDLL("libamdhip64.so")
    curve, k = h.PALLAS, 20
    n = 1 << k
    sf = co.field_of_curve(curve, "scalar")
    bases = co.generate_bases(curve, 0xC0, n)
    cols = [co.random_field(sf, 0xC1 + i, n) for i in range(3)]
    w = np.ascontiguousarray(co.generate_bases(curve, 0x77, 1)[0])
    bl = np.ascontiguousarray(cols[0][5])
    out = np.zeros(12, dtype=np.uint64)
    hd = C.c_uint64(0)
    assert lib.h2_bases_register_ex(curve, _p(bases), n, h.FORM_MONTGOMERY, 17, C.byref(hd)) == 0
    assert lib.h2_bases_set_blind_base(hd, _p(w), h.FORM_MONTGOMERY) == 0
    res = {"commit_ms_by_range": {}}
    it = [0]

    def commit():
        it[0] += 1
        rc = lib.h2_commit(hd, _p(cols[it[0] % 3]), n, None, _p(bl), h.FORM_MONTGOMERY, 0, _p(out))
        assert rc == 0
    for logc in (20, 19, 18, 17, 16):
        assert lib.h2_set_option(b"host_commit_chunk", float(1 << logc)) == 0
        for _ in range(4):
            commit()
        res["commit_ms_by_range"][f"2^{logc}"] = med(commit, 15)
    lib.h2_set_option(b"host_commit_chunk", 0.0)
    # resident lone commit for comparison
    dev = torch.device("cuda", 0)
    d_col = torch.from_numpy(cols[0].view(np.int64)).to(dev)
    d_bl = torch.from_numpy(bl.view(np.int64)).to(dev)
    d_out = torch.zeros(12, dtype=torch.int64, device=dev)

    def resident():
        lib.h2_commit_device(hd, d_col.data_ptr(), n, None, d_bl.data_ptr(), h.FORM_MONTGOMERY, 0, d_out.data_ptr(), None)
        torch.cuda.synchronize()
    for _ in range(5):
        resident()
    res["resident_lone_commit_ms"] = med(resident, 15)
    # raw PCIe
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
    hip.hipHostUnregister.argtypes = [C.c_void_p]
    d_buf = torch.empty(32 << 20, dtype=torch.uint8, device=dev)
    src = cols[1]
    for mib in (4, 32):
        sz = mib << 20
        res[f"pageable_h2d_{mib}MiB_ms"] = med(lambda: hip.hipMemcpy(d_buf.data_ptr(), src.ctypes.data, sz, 1), 9)
    t = time.perf_counter()
    rc = hip.hipHostRegister(src.ctypes.data, 32 << 20, 0)
    res["hipHostRegister_32MiB_ms"] = round((time.perf_counter() - t) * 1e3, 4)
    res["hipHostRegister_rc"] = rc
    if rc == 0:
        for mib in (4, 32):
            sz = mib << 20
            res[f"registered_h2d_{mib}MiB_ms"] = med(lambda: hip.hipMemcpy(d_buf.data_ptr(), src.ctypes.data, sz, 1), 9)
        t = time.perf_counter()
        hip.hipHostUnregister(src.ctypes.data)
        res["hipHostUnregister_32MiB_ms"] = round((time.perf_counter() - t) * 1e3, 4)
    pin = torch.empty(32 << 20, dtype=torch.uint8).pin_memory()
    for mib in (4, 32):
        sz = mib << 20
        res[f"pinned_h2d_{mib}MiB_ms"] = med(lambda: hip.hipMemcpy(d_buf.data_ptr(), pin.data_ptr(), sz, 1), 9)
    src8 = src.view(np.uint8).reshape(-1)
    pin_np = pin.numpy()
    res["host_memcpy_into_pinned_32MiB_ms"] = med(lambda: np.copyto(pin_np, src8), 5)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
