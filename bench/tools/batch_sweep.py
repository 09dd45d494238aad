"""Column-batched commits (h2_commit_batch_device, blockIdx.z = column) at 2^20 / 17-bit tables: ms per column for K columns per
call over S caller streams, 20 columns per timed region (bench.py's --steps 20 shape) and 100; K = 1 is one h2_commit_device per
column.  Also the stage times of one batched call alone (h2_profile_read: sort / accumulate / fold per launch set).
usage: batch_sweep.py [K,K,...] [S,S,...]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import halo2_amd as h
from halo2_amd.arithmetic import _p
from oracle import c_oracle as co
lib = h.lib(); lib.h2_init(0)
curve = h.PALLAS; sf = co.field_of_curve(curve, "scalar")
n = 1 << 20
Ks = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,2,4,5,8".split(","))]
Ss = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "1,2,3".split(","))]
bases = co.generate_bases(curve, 0x48414C4F32, n)
ncol = 4
cols = [co.random_field(sf, 1000 + c, n) for c in range(ncol)]
hd = C.c_uint64(0)
assert lib.h2_bases_register_ex(curve, _p(bases), n, 1, int(os.environ.get("C_BITS", "17")), C.byref(hd)) == 0
w = co.generate_bases(curve, 0x77, 1)[0]
assert lib.h2_bases_set_blind_base(hd, _p(w), 1) == 0
dev = torch.device("cuda", 0)
d_cols = [torch.from_numpy(c.view(np.int64)).to(dev) for c in cols]
d_bl = torch.from_numpy(co.random_field(sf, 0xB11D, ncol).view(np.int64)).to(dev)
d_out = torch.zeros((128, 12), dtype=torch.int64, device=dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(max(Ss))]
sps = [C.c_void_p(s.cuda_stream) for s in streams]

def run(count, K, S):
    if K <= 1:
        for i in range(count):
            c_ = i % ncol
            assert lib.h2_commit_device(hd, d_cols[c_].data_ptr(), n, None, d_bl[c_].data_ptr(), 1, 0, d_out[i % 128].data_ptr(), sps[i % S]) == 0
        return
    for j, i0 in enumerate(range(0, count, K)):
        k = min(K, count - i0)
        arr = C.c_void_p * k
        cs_ = [(i0 + q) % ncol for q in range(k)]
        rc = lib.h2_commit_batch_device(hd, arr(*[d_cols[c].data_ptr() for c in cs_]), k, n, None, arr(*[d_bl[c].data_ptr() for c in cs_]), 1, 0,
                                        arr(*[d_out[(i0 + q) % 128].data_ptr() for q in range(k)]), sps[j % S])
        assert rc == 0, lib.h2_last_error()

# SWEEP_SECONDS=t: first keep the first (K, S) running for t seconds (clock_watch.sh samples the shader clock meanwhile)
if os.environ.get("SWEEP_SECONDS"):
    run(100, Ks[0], Ss[0]); torch.cuda.synchronize()
    open(os.environ.get("SWEEP_MARK", "/tmp/sweep_started"), "w").write("1")
    t0 = time.perf_counter(); cnt = 0
    while time.perf_counter() - t0 < float(os.environ["SWEEP_SECONDS"]):
        run(100, Ks[0], Ss[0]); torch.cuda.synchronize(); cnt += 100
    dt = time.perf_counter() - t0
    print(f"sustained K={Ks[0]} S={Ss[0]}: {cnt} columns in {dt:.2f} s = {dt / cnt * 1e3:.4f} ms/col", flush=True)
ref = None
for K in Ks:
    for S in Ss:
        run(S * max(K, 1) * 2, K, S); torch.cuda.synchronize()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            run(S * max(K, 1), K, S); torch.cuda.synchronize()
        res = {}
        for count in (20, 100):
            ts = []
            for _ in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                run(count, K, S); torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / count * 1e3)
            ts.sort(); res[count] = (ts[2], ts[0])
        got = co.jac_to_affine_ints(curve, d_out[0].cpu().numpy().view(np.uint64))
        if ref is None: ref = got
        print(f"K={K} S={S}: 20 cols {res[20][0]:.4f} ms/col (min {res[20][1]:.4f}) = {n / res[20][0] / 1e3:.0f} M/s | 100 cols {res[100][0]:.4f} (min {res[100][1]:.4f}) = {n / res[100][0] / 1e3:.0f} M/s | same point: {got == ref}", flush=True)
# stage times of ONE batched call at a time on one stream
for K in Ks:
    if K <= 1: continue
    lib.h2_profile_enable(1)
    for _ in range(6):
        run(K, K, 1); torch.cuda.synchronize()
    out = []
    for name, slot in (("accumulate", 0), ("sort", 2), ("fold", 3)):
        ms, cnt = C.c_double(0), C.c_uint64(0)
        lib.h2_profile_read(slot, C.byref(ms), C.byref(cnt))
        out.append(f"{name} {ms.value / max(cnt.value, 1):.4f} ms/launch-set ({ms.value / max(cnt.value, 1) / K:.4f}/col)")
    lib.h2_profile_enable(0)
    print(f"K={K} alone: " + ", ".join(out), flush=True)
