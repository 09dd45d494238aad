"""The device-resident opening argument (halo2_amd/opening.py, every vector in HBM) against the oracle's sequential
restatement of `create_proof` (oracle/ipa.py): identical proof BYTES for the same randomness, and the oracle's
restatement of the reference verifier accepts them.  Shaped after the reference's own `test_opening_proof`
(halo2_proofs/src/poly/commitment.rs:305-379).  Runs only on a real MI355X (`-m gpu`)."""
import numpy as np
import pytest

import halo2_amd as h
from halo2_amd import fields
from halo2_amd.opening import create_proof
from halo2_amd.transcript import Blake2bWrite
from oracle import c_oracle as co
from oracle import ipa

pytestmark = pytest.mark.gpu


def _rng(sf, seed):
    ctr = [seed]

    def rng(count):
        ctr[0] += 1
        return co.random_field(sf, ctr[0], count)
    return rng


@pytest.mark.parametrize("schedule", ["collapse", "original", "paired", None])
@pytest.mark.parametrize("curve,k", [(h.PALLAS, 1), (h.PALLAS, 4), (h.PALLAS, 6), (h.VESTA, 6), (h.VESTA, 11), (h.VESTA, 13)])
def test_opening_proof_bytes_and_verification(curve, k, schedule):
    if schedule is None and k != 6:
        pytest.skip("the default schedule is 'paired' from n = 8192 on and 'original' below; its plumbing is covered once")
    n = 1 << k
    if schedule == "paired" and n < 8192:
        pytest.skip("the paired commit takes tables from 8192 points on")
    sf = fields.CURVE_FIELDS[curve][1]
    g = co.generate_bases(curve, 50 + k, n)
    w, u = co.generate_bases(curve, 60, 1)[0], co.generate_bases(curve, 61, 1)[0]
    params = h.Params.from_generators(curve, k, g, None, w, u)
    px = fields.to_limbs(range(n), sf, True)                       # commitment.rs:332-334: a_i = i
    blind = h.Blind(co.random_field(sf, 70, 1)[0])

    # --- prover on the device (commitment.rs:338-351)
    p = params.commit(px, blind, affine=True)
    tr = Blake2bWrite(curve)
    tr.write_point(p)
    x = tr.squeeze_challenge_scalar()
    v = h.eval_polynomial(px, x, sf)
    tr.write_scalar(v)
    create_proof(params, _rng(sf, 1000), tr, px, blind, x, schedule=schedule)
    ch_prover = tr.squeeze_challenge()
    proof = tr.finalize()
    assert len(proof) == 32 + 32 + 32 + 64 * k + 64

    # --- the same prover, sequential on the CPU oracle: byte-identical transcript
    p_int = co.affine_to_ints(curve, p)
    ot = ipa.Transcript(curve)
    ot.write_point(p_int)
    ox = ot.squeeze_challenge()
    assert ox == fields.from_limbs(x, sf, True)[0]
    ov = co.limbs_to_ints(co.from_mont(sf, co.eval_polynomial(sf, px, x)))[0]
    ot.write_scalar(ov)
    ipa.create_proof(curve, k, g, w, u, _rng(sf, 1000), ot, px, blind.value, x)
    assert bytes(ot.out) == proof
    assert ot.squeeze_challenge() == ch_prover

    # --- verifier (commitment.rs:353-366), oracle restatement of verifier.rs
    vt = ipa.Transcript(curve, proof)
    assert vt.read_point() == p_int
    assert vt.squeeze_challenge() == ox
    assert vt.read_scalar() == ov
    assert ipa.verify_proof(curve, k, g, w, u, vt, p_int, ox, ov)
    assert vt.squeeze_challenge() == ch_prover
    # a flipped bit in c is rejected
    bad = bytearray(proof)
    bad[-40] ^= 1
    vt = ipa.Transcript(curve, bytes(bad))
    vt.read_point(), vt.squeeze_challenge(), vt.read_scalar()
    assert not ipa.verify_proof(curve, k, g, w, u, vt, p_int, ox, ov)
    params.close()


@pytest.mark.parametrize("field", [h.FP, h.FQ])
@pytest.mark.parametrize("k,j", [(1, 0), (5, 0), (5, 2), (5, 4), (9, 5), (12, 11)])
def test_ipa_round_scalars_against_definition(field, k, j):
    """cl / cr of h2_ipa_round_scalars_device against the definition: p'[i ^ half] * prod of the challenges picked by the
    bits of h = m >> (k - j), placed by the half i falls in."""
    import torch
    from halo2_amd.arithmetic import ipa_round_scalars
    m_ = fields.MODULUS[field]
    n, blk = 1 << k, k - j
    half = 1 << (blk - 1)
    p_l = co.random_field(field, 300 + k + j, 1 << blk)
    u_l = co.random_field(field, 400 + k + j, max(j, 1))[:j]
    p_i = fields.from_limbs(p_l, field, True)
    u_i = fields.from_limbs(u_l, field, True) if j else []
    dev = torch.device("cuda:0")
    d_p = torch.from_numpy(p_l.view(np.int64)).to(dev)
    d_l = torch.full((n + 1, 4), -1, dtype=torch.int64, device=dev)
    d_r = torch.full((n + 1, 4), -1, dtype=torch.int64, device=dev)
    ipa_round_scalars(d_p, k, j, [u_l[r] for r in range(j)], field, d_l, d_r)
    got_l = fields.from_limbs(d_l[:n].cpu().numpy().view(np.uint64), field, True)
    got_r = fields.from_limbs(d_r[:n].cpu().numpy().view(np.uint64), field, True)
    for m in range(n):
        hh, i = m >> blk, m & ((1 << blk) - 1)
        s = 1
        for r in range(j):
            if (hh >> (j - 1 - r)) & 1:
                s = s * u_i[r] % m_
        v = p_i[i ^ half] * s % m_
        assert got_l[m] == (v if i < half else 0), (m, "l")
        assert got_r[m] == (0 if i < half else v), (m, "r")
    assert (d_l[n] == -1).all() and (d_r[n] == -1).all()          # the tail slot belongs to the caller


@pytest.mark.parametrize("curve,k", [(h.PALLAS, 16), (h.VESTA, 13), (h.PALLAS, 14), (h.VESTA, 15)])
def test_paired_commit_matches_two_commits(curve, k):
    """h2_commit_pair_device at k = 13 .. 16 (13- to 16-bit windows): for every shift, both outputs equal the two commits they
    stand for (the oracle's multiexp over g || u || u || w || w with the other side's scalars zeroed)."""
    import torch
    n = 1 << k
    sf = fields.CURVE_FIELDS[curve][1]
    g = co.generate_bases(curve, 91, n)
    w, u = co.generate_bases(curve, 60, 1)[0], co.generate_bases(curve, 61, 1)[0]
    params = h.Params(curve, k, g, g, w, u)
    assert params.pair_commit_supported()
    col = co.random_field(sf, 92, n + 4)
    col[::7] = 0
    basis = np.ascontiguousarray(np.concatenate([g, np.stack([u, u, w, w])]))
    d_col = torch.from_numpy(col.view(np.int64)).cuda()
    idx = np.arange(n + 4)
    for shift in (0, 1, 7, k - 1):
        side = np.where(idx < n, (idx >> shift) & 1, (idx - n) & 1)
        got = params.opening_pair_commit(d_col, shift, affine=True).cpu().numpy().view(np.uint64)
        for s_ in (0, 1):
            part = col.copy()
            part[side != s_] = 0
            want = co.jac_to_affine_ints(curve, co.best_multiexp(curve, part, basis))
            assert co.affine_to_ints(curve, got[s_]) == want
    params.close()


@pytest.mark.parametrize("curve,k", [(h.VESTA, 13), (h.PALLAS, 15)])
def test_paired_commit_sub_digit_boundaries_and_skew(curve, k):
    """Small 16-bit tables take the paired commit with 8-bit sub-digits (msm.hip, pair_subdigit_launch: every signed table digit d is
    cut again, |d| = e_0 + 256 e_1 with e_0 in [-127, 128], e_1 in [0, 128]).  Columns made of the boundary cases of that cut -- 16-bit
    windows 0x0000, 0x0001, 0x007f .. 0x0081, 0x00ff .. 0x0101, 0x7f80 / 0x7f81 (e_1 = 128 through the carry of e_0), 0x8000 (|d| = 2^15),
    0x8001 (the first digit that recodes negative and carries into the next window), 0xff7f .. 0xff81, 0xffff, in every window
    position and all at once -- and of the skewed shapes that pile entries into few of the 512 buckets (one scalar repeated, zeros,
    16-bit scalars): both outputs against the oracle's multiexp over g || u || u || w || w with the other side's scalars zeroed."""
    import random
    import torch
    n = 1 << k
    sf = fields.CURVE_FIELDS[curve][1]
    m = fields.MODULUS[sf]
    g = co.generate_bases(curve, 93, n)
    w, u = co.generate_bases(curve, 60, 1)[0], co.generate_bases(curve, 61, 1)[0]
    params = h.Params(curve, k, g, g, w, u)
    basis = np.ascontiguousarray(np.concatenate([g, np.stack([u, u, w, w])]))
    rng = random.Random(k)
    windows = [0x0000, 0x0001, 0x007F, 0x0080, 0x0081, 0x00FF, 0x0100, 0x0101, 0x7F80, 0x7F81, 0x7FFF, 0x8000, 0x8001, 0x80FF, 0xFF7F, 0xFF80,
               0xFF81, 0xFFFF]
    vals = []
    for v in windows:
        vals += [(v << (16 * pos)) % m for pos in range(16)]                                  # one window set, fifteen zero
        vals.append(sum(v << (16 * pos) for pos in range(16)) % m)                             # every window the same
        vals.append(sum((v if (pos + i_) % 2 else windows[(i_ + pos) % len(windows)]) << (16 * pos) for i_, pos in enumerate(range(16))) % m)
    vals += [0, 1, m - 1, m - 2, (m - 1) // 2, (m + 1) // 2]
    idx = np.arange(n + 4)

    def check(col, shifts):
        d_col = torch.from_numpy(np.ascontiguousarray(col).view(np.int64)).cuda()
        for shift in shifts:
            side = np.where(idx < n, (idx >> shift) & 1, (idx - n) & 1)
            got = params.opening_pair_commit(d_col, shift, affine=False).cpu().numpy().view(np.uint64)
            for s_ in (0, 1):
                part = col.copy()
                part[side != s_] = 0
                assert co.jac_to_affine_ints(curve, got[s_]) == co.jac_to_affine_ints(curve, co.best_multiexp(curve, part, basis)), (shift, s_)
    col = co.random_field(sf, 94, n + 4)
    where = rng.sample(range(n + 4), len(vals))
    col[where] = fields.to_limbs(vals, sf, True)
    check(col, (0, 3, k - 1))
    alone = np.zeros_like(col)
    alone[where] = col[where]
    check(alone, (0, k - 1))                                                                    # the boundary scalars alone: no cancellation can hide a weight
    same = np.tile(co.random_field(sf, 95, 1), (n + 4, 1))                                     # one scalar everywhere: 32 buckets hold everything
    check(same, (1,))
    small = fields.to_limbs([(i * 2654435761) & 0xFFFF for i in range(n + 4)], sf, True)       # one window in use
    check(small, (2,))
    check(np.zeros_like(col), (0,))                                                             # nothing at all: two identities
    params.close()


def test_opening_proof_full_size_schedules_agree_and_verify():
    """k = 20 (BASELINE's size): the two schedules -- independent device algorithms for L_j, R_j -- write identical proof
    bytes, and the oracle's restatement of the reference verifier (one 2^20 multiexp on the host) accepts them."""
    curve, k = h.VESTA, 20
    n = 1 << k
    sf = fields.CURVE_FIELDS[curve][1]
    g = co.generate_bases(curve, 77, n)
    w, u = co.generate_bases(curve, 60, 1)[0], co.generate_bases(curve, 61, 1)[0]
    params = h.Params(curve, k, g, g, w, u)                        # g_lagrange is not used by the opening argument
    px = co.random_field(sf, 78, n)
    blind = h.Blind(co.random_field(sf, 70, 1)[0])
    p = params.commit(px, blind, affine=True)
    proofs = []
    assert params.default_hybrid_rounds(True) == 6
    # "paired" alone moves to the collapsed generators after 6 rounds (the default at k = 20); 0 keeps every round on the original ones
    for schedule, hybrid in (("original", None), ("collapse", None), ("paired", 0), ("paired", None), ("paired", 5), ("paired", 9)):
        tr = Blake2bWrite(curve)
        tr.write_point(p)
        x = tr.squeeze_challenge_scalar()
        v = h.eval_polynomial(px, x, sf)
        tr.write_scalar(v)
        create_proof(params, _rng(sf, 2000), tr, px, blind, x, schedule=schedule, hybrid_rounds=hybrid)
        proofs.append(tr.finalize())
    assert all(pr == proofs[0] for pr in proofs) and len(proofs[0]) == 32 + 32 + 32 + 64 * k + 64
    p_int = co.affine_to_ints(curve, p)
    vt = ipa.Transcript(curve, proofs[0])
    assert vt.read_point() == p_int
    ox = vt.squeeze_challenge()
    ov = vt.read_scalar()
    assert ox == fields.from_limbs(x, sf, True)[0] and ov == fields.from_limbs(v, sf, True)[0]
    assert ipa.verify_proof(curve, k, g, w, u, vt, p_int, ox, ov)
    params.close()


@pytest.mark.parametrize("curve,k,rounds", [(h.PALLAS, 16, 1), (h.VESTA, 16, 3), (h.PALLAS, 16, 6), (h.VESTA, 17, 12)])
def test_collapsed_generators_against_definition(curve, k, rounds):
    """h2_ipa_collapsed_generators_device: G'[i] = sum_h s(h) G[i + h 2^(k - rounds)] with s(h) the product of the challenges
    picked by the bits of h (bit rounds-1-r <-> u_r), checked at sampled i against the oracle's multiexp; and a table registered
    from the device copy (h2_bases_register_device) commits like one registered from the host."""
    import ctypes as C
    import torch
    from halo2_amd._lib import FORM_MONTGOMERY, check, lib
    n, nj = 1 << k, 1 << (k - rounds)
    sf = fields.CURVE_FIELDS[curve][1]
    m_ = fields.MODULUS[sf]
    g = co.generate_bases(curve, 93, n)
    w, u = co.generate_bases(curve, 60, 1)[0], co.generate_bases(curve, 61, 1)[0]
    params = h.Params(curve, k, g, g, w, u)
    assert params.pair_commit_supported()
    handle = params._opening_basis(True)
    ch = co.random_field(sf, 94, rounds)
    if rounds >= 3:
        ch[1] = fields.scalar_limbs(m_ - 1, sf, True)              # every 16-bit digit at its extreme
        ch[2] = fields.scalar_limbs(0x8000800080008000, sf, True)
    u_i = fields.from_limbs(ch, sf, True)
    d_out = torch.empty((nj, 8), dtype=torch.int64, device="cuda:0")
    check(lib().h2_ipa_collapsed_generators_device(handle, k, rounds, ch.ctypes.data_as(C.POINTER(C.c_uint64)), FORM_MONTGOMERY,
                                                   d_out.data_ptr(), None), "h2_ipa_collapsed_generators_device")
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().view(np.uint64)
    s_ = []
    for hh in range(1 << rounds):
        v = 1
        for r in range(rounds):
            if (hh >> (rounds - 1 - r)) & 1:
                v = v * u_i[r] % m_
        s_.append(v)
    s_l = fields.to_limbs(s_, sf, True)
    for i in sorted({0, 1, nj - 1, nj // 2, (12345 * 7) % nj}):
        pts = np.ascontiguousarray(g[i::nj])
        assert pts.shape[0] == 1 << rounds
        want = co.jac_to_affine_ints(curve, co.best_multiexp(curve, s_l, pts))
        assert co.affine_to_ints(curve, got[i]) == want, i
    # registration from device memory
    hj = C.c_uint64(0)
    check(lib().h2_bases_register_device(curve, d_out.data_ptr(), nj, FORM_MONTGOMERY, C.byref(hj)), "h2_bases_register_device")
    col = co.random_field(sf, 95, nj)
    d_col = torch.from_numpy(col.view(np.int64)).cuda()
    out = torch.empty(8, dtype=torch.int64, device="cuda:0")
    check(lib().h2_commit_device(hj, d_col.data_ptr(), nj, None, None, FORM_MONTGOMERY, 1, out.data_ptr(), None), "h2_commit_device")
    torch.cuda.synchronize()
    want = co.jac_to_affine_ints(curve, co.best_multiexp(curve, col, np.ascontiguousarray(got)))
    assert co.affine_to_ints(curve, out.cpu().numpy().view(np.uint64)) == want
    lib().h2_bases_free(hj)
    params.close()


@pytest.mark.parametrize("curve,k,hybrid", [(h.VESTA, 16, 1), (h.PALLAS, 16, 3), (h.VESTA, 16, 9), (h.PALLAS, 17, None)])
def test_opening_hybrid_schedule_same_bytes(curve, k, hybrid):
    """The opening argument that moves to the collapsed generators after `hybrid` rounds writes the bytes of the one that stays on
    the original generators, and the restated verifier accepts them."""
    n = 1 << k
    sf = fields.CURVE_FIELDS[curve][1]
    g = co.generate_bases(curve, 96, n)
    w, u = co.generate_bases(curve, 60, 1)[0], co.generate_bases(curve, 61, 1)[0]
    params = h.Params(curve, k, g, g, w, u)
    px = co.random_field(sf, 97, n)
    blind = h.Blind(co.random_field(sf, 70, 1)[0])
    p = params.commit(px, blind, affine=True)
    proofs = []
    for schedule, hy in (("original", None), ("paired", hybrid)):
        tr = Blake2bWrite(curve)
        tr.write_point(p)
        x = tr.squeeze_challenge_scalar()
        v = h.eval_polynomial(px, x, sf)
        tr.write_scalar(v)
        create_proof(params, _rng(sf, 3000), tr, px, blind, x, schedule=schedule, hybrid_rounds=hy)
        proofs.append(tr.finalize())
    assert proofs[0] == proofs[1]
    p_int = co.affine_to_ints(curve, p)
    vt = ipa.Transcript(curve, proofs[1])
    assert vt.read_point() == p_int
    ox = vt.squeeze_challenge()
    ov = vt.read_scalar()
    assert ipa.verify_proof(curve, k, g, w, u, vt, p_int, ox, ov)
    params.close()


@pytest.mark.parametrize("curve,k,schedule", [(h.VESTA, 16, None), (h.PALLAS, 16, "collapse"), (h.VESTA, 16, "original"), (h.VESTA, 20, None),
                                              (h.PALLAS, 20, "collapse")])
def test_opening_proof_bytes_at_size_against_the_c_restatement(curve, k, schedule):
    """Proof-level parity at the sizes the bench quotes (missing 4): the device opening argument at k = 16 and
    k = 20 -- the library's default schedule (paired commits over the original generators, then the collapsed generators read
    off the table) and the reference's own schedule (`collapse`) -- writes the BYTES of the sequential restatement of
    `commitment::create_proof` (poly/commitment/prover.rs:27-151), whose multiexps, inner products, folds and generator
    collapse are the C oracle's (oracle/ipa.py over oracle/h2_oracle.c), for the same polynomial, blind and randomness."""
    n = 1 << k
    sf = fields.CURVE_FIELDS[curve][1]
    g = co.generate_bases(curve, 0x1600 + k, n)
    w, u = co.generate_bases(curve, 60, 1)[0], co.generate_bases(curve, 61, 1)[0]
    params = h.Params(curve, k, g, g, w, u)                       # g_lagrange plays no part in the opening argument
    px = co.random_field(sf, 0x1601 + k, n)
    blind = h.Blind(co.random_field(sf, 70, 1)[0])
    p = params.commit(px, blind, affine=True)
    tr = Blake2bWrite(curve)
    tr.write_point(p)
    x = tr.squeeze_challenge_scalar()
    v = h.eval_polynomial(px, x, sf)
    tr.write_scalar(v)
    create_proof(params, _rng(sf, 5000), tr, px, blind, x, schedule=schedule)
    proof = tr.finalize()
    p_int = co.affine_to_ints(curve, p)
    assert p_int == co.jac_to_affine_ints(curve, co.commit(curve, g, w, px, blind.value))
    ot = ipa.Transcript(curve)
    ot.write_point(p_int)
    assert ot.squeeze_challenge() == fields.from_limbs(x, sf, True)[0]
    ot.write_scalar(co.limbs_to_ints(co.from_mont(sf, co.eval_polynomial(sf, px, x)))[0])
    ipa.create_proof(curve, k, g, w, u, _rng(sf, 5000), ot, px, blind.value, x)
    assert bytes(ot.out) == proof
    params.close()


@pytest.mark.parametrize("k", [5, 13, 16])
def test_native_host_mirror_opening_same_bytes(k):
    """The C++ host mirror's create_proof + Blake2bWrite (halo2_amd/host/halo2_host.hpp, plain g++ over the C ABI: h2_commit,
    h2_scale_add, h2_powers, h2_ipa_rounds) writes the bytes of the Python mirror for the same Params::new(k), polynomial and
    randomness (k = 16 switches to the collapsed generators inside the library)."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "host_mirror_check")
    if not os.path.exists(exe):
        pytest.skip(f"{exe} not built (run __graft_entry__.build())")
    out = subprocess.run([exe, "opening", str(k)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    native = bytes.fromhex(out.stdout.strip().splitlines()[-1])
    curve = h.VESTA
    sf = fields.CURVE_FIELDS[curve][1]
    n = 1 << k
    params = h.Params.new(curve, k)
    px = fields.to_limbs(range(n), sf, True)
    blind = h.Blind(fields.scalar_limbs(7, sf, True))
    ctr = [0]

    def rng(count):
        vals = [((ctr[0] + i + 1) * 0x9E3779B97F4A7C15 + 1) % (1 << 64) for i in range(count)]
        ctr[0] += count
        return fields.to_limbs(vals, sf, True)
    tr = Blake2bWrite(curve)
    tr.write_point(params.commit(px, blind, affine=True))
    x = tr.squeeze_challenge_scalar()
    tr.write_scalar(h.eval_polynomial(px, x, sf))
    create_proof(params, rng, tr, px, blind, x)
    assert tr.finalize() == native
    params.close()


@pytest.mark.parametrize("curve,k", [(h.PALLAS, 1), (h.VESTA, 6), (h.PALLAS, 13), (h.VESTA, 16)])
def test_whole_argument_entry_points_write_the_oracles_bytes(curve, k):
    """`commitment::create_proof` as ONE native call (h2_open from host vectors; with p_poly resident h2_open_device_host_s, which moves the host rng's
    s_poly across in quarters and commits each as it lands from k = 16 on -- the k = 16 case here --, or h2_open_device when the rng draws on the device) against the
    sequential restatement of the reference prover, and against the step-by-step form it replaces (native=False: the steps before
    the loop from Python, h2_ipa_rounds_device for the loop) -- four routes, one byte string.  k = 1: a single round, no table
    pair; 6: the two-commit rounds; 13: the paired rounds; 16: the switch to the collapsed generators inside the call."""
    import torch
    n = 1 << k
    sf = fields.CURVE_FIELDS[curve][1]
    g = co.generate_bases(curve, 150 + k, n)
    w, u = co.generate_bases(curve, 60, 1)[0], co.generate_bases(curve, 61, 1)[0]
    params = h.Params.from_generators(curve, k, g, None, w, u)
    px = co.random_field(sf, 170 + k, n)
    blind = h.Blind(co.random_field(sf, 171, 1)[0])
    x = co.random_field(sf, 172, 1)[0]
    d_px = torch.from_numpy(px.view(np.int64)).to("cuda:0")
    proofs = []
    for resident in (False, True):
        for native in (True, False):
            tr = Blake2bWrite(curve)
            create_proof(params, _rng(sf, 9000), tr, d_px.clone() if resident else px.copy(), blind, x, native=native)
            proofs.append(tr.finalize())
    assert torch.equal(d_px.cpu(), torch.from_numpy(px.view(np.int64)))          # p_poly is read only
    assert proofs[0] == proofs[1] == proofs[2] == proofs[3]
    ot = ipa.Transcript(curve)
    ipa.create_proof(curve, k, g, w, u, _rng(sf, 9000), ot, px, blind.value, x)
    assert bytes(ot.out) == proofs[0]
    params.close()


def test_whole_argument_refuses_before_touching_the_transcript():
    """What h2_open can refuse it refuses before the commitment to s_poly reaches the caller's transcript: a g table without
    Params::w installed, tables of the wrong size, a missing argument."""
    import ctypes as C
    from halo2_amd._lib import IPA_SQUEEZE_FN, IPA_WRITE_POINT_FN, IPA_SWITCH_DEFAULT, lib
    from halo2_amd.arithmetic import _p
    curve, k = h.VESTA, 6
    n = 1 << k
    sf = fields.CURVE_FIELDS[curve][1]
    g = co.generate_bases(curve, 180, n)
    w, u = co.generate_bases(curve, 60, 1)[0], co.generate_bases(curve, 61, 1)[0]
    params = h.Params.from_generators(curve, k, g, None, w, u)
    calls = []
    cb_w = IPA_WRITE_POINT_FN(lambda _u, _xy: calls.append("w") or 0)
    cb_s = IPA_SQUEEZE_FN(lambda _u, _out: calls.append("s") or 0)
    px, s = co.random_field(sf, 181, n), co.random_field(sf, 182, n)
    one = co.random_field(sf, 183, 1)[0]
    rands = co.random_field(sf, 184, 2 * k)
    uw = np.ascontiguousarray(np.stack([params.u, params.w]))
    c, f = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
    bare = C.c_uint64(0)                                     # the same generators, no blind base installed
    assert lib().h2_bases_register(curve, _p(np.ascontiguousarray(g)), n, 1, C.byref(bare)) == 0
    basis = params._opening_basis(False)

    def call(g_handle, open_handle, paired, p_ptr=_p(px)):
        return lib().h2_open(curve, k, g_handle, open_handle, paired, IPA_SWITCH_DEFAULT, _p(uw), p_ptr, _p(one), _p(one), _p(s), _p(one), _p(rands),
                             cb_w, cb_s, None, _p(c), _p(f))
    assert call(bare, basis, 0) != 0                         # no Params::w on the g table
    assert call(params._h_g, params._h_g, 0) != 0            # the opening basis must be g || u || w
    assert call(params._h_g, basis, 1) != 0                  # ... or g || u || u || w || w when paired
    assert call(params._h_g, basis, 0, None) != 0            # p_poly missing
    assert call(C.c_uint64(0xDEAD), basis, 0) != 0           # not a handle
    import torch
    d_same = torch.from_numpy(px.view(np.int64)).cuda()
    assert lib().h2_open_device(curve, k, params._h_g, basis, 0, IPA_SWITCH_DEFAULT, _p(uw), d_same.data_ptr(), _p(one), _p(one), d_same.data_ptr(), _p(one),
                                _p(rands), cb_w, cb_s, None, _p(c), _p(f), None) != 0       # p_poly and s_poly must not be one buffer
    assert lib().h2_open_device_host_s(curve, k, bare, basis, 0, IPA_SWITCH_DEFAULT, _p(uw), d_same.data_ptr(), _p(one), _p(one), _p(s), _p(one), _p(rands),
                                       cb_w, cb_s, None, _p(c), _p(f), None) != 0                 # resident p_poly, host s_poly: the same refusals
    assert lib().h2_open_device_host_s(curve, k, params._h_g, basis, 0, IPA_SWITCH_DEFAULT, _p(uw), d_same.data_ptr(), _p(one), _p(one), None, _p(one), _p(rands),
                                       cb_w, cb_s, None, _p(c), _p(f), None) != 0
    assert calls == []
    lib().h2_bases_free(bare)
    params.close()
