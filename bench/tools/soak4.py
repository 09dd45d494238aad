"""Randomised soak of the round-4 paths against the C restatement (not part of the suite: its time is spent in the CPU oracle):
  * h2_commit_batch_device in its column-batched form: tables of random size 2^10 .. 2^17 (window width as Params would register
    it, or forced 13 / 16 / 17 bits), 2 .. 11 columns per call (one or two launch groups), random column patterns side by side
    (dense, 90 % zeros, one repeated scalar, < 2^16, half repeated, 2^128 - 1, all zero), with / without blinds, full or prefix length,
    identity and duplicate bases -- every output against orc_commit / orc_best_multiexp;
  * best_fft at random sizes 2^1 .. 2^21 on both fields with random (non-root) omegas, plus the fused ifft, elementwise.
python bench/tools/soak4.py [seconds]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import halo2_amd as h
from halo2_amd import fields
from halo2_amd.arithmetic import _p
from oracle import c_oracle as co

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(20260925)
lib = h.lib(); lib.h2_init(0)
dev = torch.device("cuda", 0)
t_end = time.time() + budget
commits = ffts = fails = 0


def pattern(sf, n, seed):
    col = co.random_field(sf, seed, n)
    kind = int(rng.integers(0, 8))
    if kind == 1: col[rng.random(n) < 0.9] = 0
    elif kind == 2: col[:] = col[0]
    elif kind == 3: col[:, 1:] = 0; col = co.to_mont(sf, col & 0xFFFF)
    elif kind == 4: col[rng.random(n) < 0.5] = col[0]
    elif kind == 5: col[rng.integers(0, n, size=max(1, n // 50))] = 0
    elif kind == 6: col = co.to_mont(sf, np.tile(np.array([[0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF, 0, 0]], dtype=np.uint64), (n, 1)))
    elif kind == 7: col[:] = 0                     # no entries at all (an empty stretch of a joined batch)
    return np.ascontiguousarray(col), kind


while time.time() < t_end:
    # ---- batched commits
    curve = int(rng.integers(0, 2))
    sf = co.field_of_curve(curve, "scalar")
    logn = int(rng.integers(10, 18))
    n = int(rng.integers(1 << logn, (2 << logn)))
    g = co.generate_bases(curve, int(rng.integers(1, 1 << 30)), n)
    if rng.random() < 0.3: g[rng.integers(0, n)] = 0
    if rng.random() < 0.3: g[1] = g[0]
    bits = int(rng.choice([0, 0, 13, 16, 17]))
    if bits == 17 and n < (1 << 16): bits = 16
    hd = C.c_uint64(0)
    rc = lib.h2_bases_register_ex(curve, _p(g), n, 1, bits, C.byref(hd))
    if rc != 0:
        assert lib.h2_bases_register_ex(curve, _p(g), n, 1, 0, C.byref(hd)) == 0
    w = co.generate_bases(curve, int(rng.integers(1, 1 << 30)), 1)[0]
    assert lib.h2_bases_set_blind_base(hd, _p(w), 1) == 0
    for _ in range(2):
        count = int(rng.integers(2, 12))
        cols, kinds = zip(*[pattern(sf, n, int(rng.integers(1, 1 << 30))) for _ in range(count)])
        used = n if rng.random() < 0.6 else int(rng.integers(1, n + 1))
        with_blind = rng.random() < 0.6
        blinds = co.random_field(sf, int(rng.integers(1, 1 << 30)), count)
        d_cols = [torch.from_numpy(c.view(np.int64)).to(dev) for c in cols]
        d_bl = torch.from_numpy(blinds.view(np.int64)).to(dev)
        d_out = torch.zeros((count, 12), dtype=torch.int64, device=dev)
        arr = C.c_void_p * count
        rc = lib.h2_commit_batch_device(hd, arr(*[c.data_ptr() for c in d_cols]), count, used, None,
                                        arr(*[d_bl[i].data_ptr() for i in range(count)]) if with_blind else None, 1, 0,
                                        arr(*[d_out[i].data_ptr() for i in range(count)]), None)
        assert rc == 0, lib.h2_last_error()
        torch.cuda.synchronize()
        got = d_out.cpu().numpy().view(np.uint64)
        for i in range(count):
            want = (co.commit(curve, np.ascontiguousarray(g[:used]), w, np.ascontiguousarray(cols[i][:used]), blinds[i]) if with_blind
                    else co.best_multiexp(curve, cols[i][:used], g[:used]))
            commits += 1
            if co.jac_to_affine_ints(curve, got[i]) != co.jac_to_affine_ints(curve, want):
                fails += 1
                print("MISMATCH commit: curve", curve, "n", n, "bits", bits, "used", used, "count", count, "column", i, "pattern", kinds[i], "blind", with_blind, flush=True)
    lib.h2_bases_free(hd)
    # ---- transforms
    for _ in range(3):
        field = int(rng.integers(0, 2))
        L = int(rng.integers(1, 22))
        a = co.random_field(field, int(rng.integers(1, 1 << 30)), 1 << L)
        omega = co.random_field(field, int(rng.integers(1, 1 << 30)), 1)[0]
        ffts += 1
        if not np.array_equal(h.best_fft(a.copy(), omega, L, field), co.best_fft(field, a, omega, L)):
            fails += 1
            print("MISMATCH best_fft: field", field, "log_n", L, flush=True)
        div = co.random_field(field, int(rng.integers(1, 1 << 30)), 1)[0]
        b = a.copy()
        assert lib.h2_ifft(field, _p(b), L, _p(omega), _p(div), 1) == 0
        ffts += 1
        if not np.array_equal(b, co.ifft(field, a, omega, L, div)):
            fails += 1
            print("MISMATCH ifft: field", field, "log_n", L, flush=True)
print(f"soak4: {commits} batched column commits and {ffts} transforms checked, {fails} mismatches")
sys.exit(1 if fails else 0)
