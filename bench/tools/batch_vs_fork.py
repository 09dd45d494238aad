"""h2_commit_batch_device, the column-batched form (one launch set, blockIdx.z = column) against the forked form (one commit per
column over three internal streams; H2_BATCH_COLS=1), ms per column for `count` columns of 2^k scalars on tables registered as
Params does (h2_commit_column_window_bits).  Run twice: once plain, once with H2_BATCH_COLS=1."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import halo2_amd as h
from halo2_amd.arithmetic import _p
from oracle import c_oracle as co
lib = h.lib(); lib.h2_init(0)
curve = h.VESTA; sf = co.field_of_curve(curve, "scalar")
dev = torch.device("cuda", 0)
mode = "forked (H2_BATCH_COLS=1)" if os.environ.get("H2_BATCH_COLS") == "1" else "batched"
for k in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "11,12,13,14,16,18,20".split(","))]:
    n = 1 << k
    bases = co.generate_bases(curve, 7 + k, n)
    hd = C.c_uint64(0)
    assert lib.h2_bases_register_ex(curve, _p(bases), n, 1, int(lib.h2_commit_column_window_bits(n)), C.byref(hd)) == 0
    w = co.generate_bases(curve, 0x77, 1)[0]
    assert lib.h2_bases_set_blind_base(hd, _p(w), 1) == 0
    cols = [torch.from_numpy(co.random_field(sf, 100 + c, n).view(np.int64)).to(dev) for c in range(8)]
    d_bl = torch.from_numpy(co.random_field(sf, 0xB11D, 8).view(np.int64)).to(dev)
    d_out = torch.zeros((8, 12), dtype=torch.int64, device=dev)
    for count in (2, 3, 8):
        arr = C.c_void_p * count
        sc, bl, outs = arr(*[cols[i].data_ptr() for i in range(count)]), arr(*[d_bl[i].data_ptr() for i in range(count)]), arr(*[d_out[i].data_ptr() for i in range(count)])
        def run():
            assert lib.h2_commit_batch_device(hd, sc, count, n, None, bl, 1, 0, outs, None) == 0
        for _ in range(5): run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            for _ in range(5): run()
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 5 * 1e3)
        ts.sort()
        print(f"{mode}: 2^{k} x {count} columns: {ts[3]:.4f} ms per call, {ts[3] / count:.4f} ms per column", flush=True)
    lib.h2_bases_free(hd)
