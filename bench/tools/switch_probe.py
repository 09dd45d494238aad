#!/usr/bin/env python3
"""The pieces of the opening argument's switch to collapsed generators at k = 20, J = 6, each alone with a synchronise around it:
the read-out of G'_J off the table (h2_ipa_collapsed_generators_device), the registration of the 2^(k-J) + 4 points
(h2_bases_register_device) and the release of that table (h2_bases_free)."""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    import torch
    import halo2_amd as h
    from halo2_amd import fields
    from halo2_amd._lib import lib
    from oracle import c_oracle as co          # input generation only
    k, curve = int(os.environ.get("K", "20")), 1
    n = 1 << k
    sf = fields.CURVE_FIELDS[curve][1]
    dev = torch.device("cuda:0")
    g = co.generate_bases(curve, 1, n)
    w, u = co.generate_bases(curve, 2, 1)[0], co.generate_bases(curve, 3, 1)[0]
    params = h.Params(curve, k, g, g, w, u)
    basis = params._opening_basis(True)
    L = lib()
    res = {"k": k}
    for J in [int(x) for x in os.environ.get("JS", "3,4,5,6,7,8").split(",")]:
        nj = 1 << (k - J)
        ch = np.ascontiguousarray(co.random_field(sf, 77, J))
        d_g = torch.zeros((nj + 4, 8), dtype=torch.int64, device=dev)
        ts = {"readout": [], "register": [], "free": []}
        for rep in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rc = L.h2_ipa_collapsed_generators_device(basis, k, J, ch.ctypes.data_as(C.POINTER(C.c_uint64)), 1, d_g.data_ptr(), None)
            assert rc == 0, L.h2_last_error()
            torch.cuda.synchronize(); t1 = time.perf_counter()
            d_g[nj:] = d_g[:4]
            torch.cuda.synchronize(); t1b = time.perf_counter()
            hj = C.c_uint64(0)
            assert L.h2_bases_register_device(curve, d_g.data_ptr(), nj + 4, 1, C.byref(hj)) == 0
            torch.cuda.synchronize(); t2 = time.perf_counter()
            assert L.h2_bases_free(hj) == 0
            torch.cuda.synchronize(); t3 = time.perf_counter()
            ts["readout"].append(t1 - t0); ts["register"].append(t2 - t1b); ts["free"].append(t3 - t2)
        res[f"J={J}"] = {k_: round(sorted(v)[len(v) // 2] * 1e3, 3) for k_, v in ts.items()}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
