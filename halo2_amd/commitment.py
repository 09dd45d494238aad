"""`Params` (halo2_proofs/src/poly/commitment.rs:26-205) over the C ABI: `g` and `g_lagrange` are
registered once and stay in HBM; `commit` / `commit_lagrange` (:119-150) ship only the column.

`Params.new(curve, k)` derives the generators as `Params::new` does (commitment.rs:38-114): pasta_curves'
hash-to-curve (h2_hash_to_curve, csrc/h2c.hip) and the point iFFT, both on the device; `from_generators` takes
caller-supplied generators, which is also what `Params::read` (:184) amounts to."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import fields
from ._lib import FORM_MONTGOMERY, OUT_AFFINE, OUT_JACOBIAN, check, lib
from .arithmetic import _is_torch, _np, _p, _stream_ptr


class Blind:
    """Blind<F> (commitment.rs:208-256): newtype around the blinding scalar, default 1."""

    def __init__(self, value: np.ndarray | None = None, field: int | None = None):
        self.value = fields.scalar_limbs(1, field, True) if value is None else np.ascontiguousarray(value, dtype=np.uint64)


def lagrange_basis(g, curve: int, k: int, form: int = FORM_MONTGOMERY) -> np.ndarray:
    """The point FFT of `Params::new` (commitment.rs:77-100): g_lagrange from g, on the device."""
    g = _np(g, 8)
    if g.shape[0] != 1 << k:
        raise ValueError("lagrange_basis: need 2^k generators")
    out = np.empty_like(g)
    check(lib().h2_lagrange_basis(curve, _p(g), _p(out), k, form), "h2_lagrange_basis")
    return out


def points_to_bytes(points, curve: int, form: int = FORM_MONTGOMERY) -> bytes:
    """pasta_curves `to_bytes` for n affine points (n, 8): 32 bytes each, x little-endian | parity(y) << 255."""
    pts = _np(np.asarray(points).reshape(-1, 8), 8)
    out = np.zeros(pts.shape[0] * 32, dtype=np.uint8)
    check(lib().h2_points_compress(curve, _p(pts), pts.shape[0], form, out.ctypes.data_as(C.POINTER(C.c_uint8))), "h2_points_compress")
    return out.tobytes()


def points_from_bytes(raw: bytes, curve: int, form: int = FORM_MONTGOMERY) -> np.ndarray:
    """`from_bytes` for len(raw) / 32 points -> (n, 8) affine limbs.  ValueError on an invalid encoding (the reference
    returns an io::Error from Params::read, commitment.rs:193-198)."""
    if len(raw) % 32:
        raise ValueError("points_from_bytes: length is not a multiple of 32")
    buf = np.frombuffer(raw, dtype=np.uint8).copy()
    n = buf.shape[0] // 32
    out = np.zeros((n, 8), dtype=np.uint64)
    check(lib().h2_points_decompress(curve, buf.ctypes.data_as(C.POINTER(C.c_uint8)), n, form, _p(out)), "h2_points_decompress")
    return out


def hash_to_curve(curve: int, domain_prefix: str, messages, form: int = FORM_MONTGOMERY) -> np.ndarray:
    """`C::CurveExt::hash_to_curve(domain_prefix)` applied to each message (bytes objects of equal length, or an (n, len) uint8
    array) -> (n, 8) affine limbs.  pasta_curves' map (BLAKE2b XMD, simplified SWU, 3-isogeny), computed on the device."""
    if isinstance(messages, np.ndarray):
        msgs = np.ascontiguousarray(messages, dtype=np.uint8)
    else:
        messages = list(messages)
        if len({len(m_) for m_ in messages}) > 1:
            raise ValueError("hash_to_curve: messages must have equal length")
        msgs = np.frombuffer(b"".join(messages), dtype=np.uint8).reshape(len(messages), -1 if messages and len(messages[0]) else 0).copy()
    n, mlen = msgs.shape[0], (msgs.shape[1] if msgs.ndim == 2 else 0)
    out = np.zeros((n, 8), dtype=np.uint64)
    check(lib().h2_hash_to_curve(curve, domain_prefix.encode(), msgs.ctypes.data_as(C.c_void_p), mlen, n, form, _p(out)), "h2_hash_to_curve")
    return out


def _transcript_callbacks(transcript):
    """write_point / squeeze of the opening argument's native calls over `transcript`: (cb_w, cb_s, user, failures).  A transcript of
    the library (halo2_amd.transcript, a `handle` attribute) is called natively, no Python in between; any other object through
    ctypes callbacks whose exceptions are parked in `failures` (an exception must not unwind through the C frames)."""
    from ._lib import IPA_SQUEEZE_FN, IPA_WRITE_POINT_FN
    failure = []
    if getattr(transcript, "handle", 0):
        return (C.cast(lib().h2_transcript_cb_write_point, IPA_WRITE_POINT_FN), C.cast(lib().h2_transcript_cb_squeeze, IPA_SQUEEZE_FN),
                C.c_void_p(transcript.handle), failure)

    def write_point(_user, xy):
        try:
            transcript.write_point(np.ctypeslib.as_array(xy, shape=(8,)).copy())
            return 0
        except Exception as e:
            failure.append(e)
            return 1

    def squeeze(_user, out):
        try:
            np.ctypeslib.as_array(out, shape=(4,))[:] = np.asarray(transcript.squeeze_challenge_scalar(), dtype=np.uint64).reshape(4)
            return 0
        except Exception as e:
            failure.append(e)
            return 1
    return IPA_WRITE_POINT_FN(write_point), IPA_SQUEEZE_FN(squeeze), None, failure


class Params:
    def __init__(self, curve: int, k: int, g, g_lagrange, w, u):
        self.curve, self.k, self.n = curve, k, 1 << k
        self.g = _np(g, 8)
        self.g_lagrange = _np(g_lagrange, 8)
        if self.g.shape[0] != self.n or self.g_lagrange.shape[0] != self.n:
            raise ValueError("Params: g / g_lagrange must have 2^k points")
        self.w = np.ascontiguousarray(w, dtype=np.uint64).reshape(8)
        self.u = np.ascontiguousarray(u, dtype=np.uint64).reshape(8)
        self._h_g = C.c_uint64(0)
        self._h_gl = C.c_uint64(0)
        self.device_index = int(lib().h2_current_device())     # the registered tables live on the device current now
        # tables that only serve column commits take the width that is fastest for independent commits (17 bits from 2^18 on)
        wb = int(lib().h2_commit_column_window_bits(self.n))
        check(lib().h2_bases_register_ex(curve, _p(self.g), self.n, FORM_MONTGOMERY, wb, C.byref(self._h_g)), "h2_bases_register_ex")
        check(lib().h2_bases_register_ex(curve, _p(self.g_lagrange), self.n, FORM_MONTGOMERY, wb, C.byref(self._h_gl)),
              "h2_bases_register_ex")
        # `w` is a field of Params (commitment.rs:26-33): installed once per table, commits then pass only their blind scalar
        for h_ in (self._h_g, self._h_gl):
            check(lib().h2_bases_set_blind_base(h_, _p(self.w), FORM_MONTGOMERY), "h2_bases_set_blind_base")
        self._h_gu = C.c_uint64(0)        # g || u, registered on the first opening argument (opening.py)
        self._h_pair = C.c_uint64(0)      # g || u || u || w || w for the paired L_j / R_j commits
        self._h_guw = C.c_uint64(0)       # g || u || w for the two-commit rounds of small arguments

    @classmethod
    def new(cls, curve: int, k: int) -> "Params":
        """`Params::new(k)` (commitment.rs:38-114): g_i = hash_to_curve("Halo2-Parameters")({0, i as LE u32}), g_lagrange by the
        point iFFT, w = hasher({1}), u = hasher({2}) -- all on the device.  Bit-exact with the reference's generators (the
        verifying key pinned in tests/plonk_api.rs:958-981 is reproduced from Params.new(5))."""
        if not 0 <= k < 32:
            raise ValueError("Params.new: k out of range")                   # assert!(k < 32), commitment.rs:41
        n = 1 << k
        msgs = np.zeros((n, 5), dtype=np.uint8)
        msgs[:, 1:5] = np.arange(n, dtype="<u4").view(np.uint8).reshape(n, 4)
        g = hash_to_curve(curve, "Halo2-Parameters", msgs)
        w = hash_to_curve(curve, "Halo2-Parameters", [b"\x01"])[0]
        u = hash_to_curve(curve, "Halo2-Parameters", [b"\x02"])[0]
        return cls.from_generators(curve, k, g, None, w, u)

    @classmethod
    def from_generators(cls, curve: int, k: int, g, g_lagrange, w, u) -> "Params":
        """g_lagrange=None: derive it from g with the device point FFT, as Params::new does (:77-100)."""
        if g_lagrange is None:
            g_lagrange = lagrange_basis(g, curve, k)
        return cls(curve, k, g, g_lagrange, w, u)

    def write(self, writer) -> None:
        """Params::write (commitment.rs:169-181): k (u32 LE), g, g_lagrange, w, u as compressed points."""
        writer.write(int(self.k).to_bytes(4, "little"))
        writer.write(points_to_bytes(self.g, self.curve))
        writer.write(points_to_bytes(self.g_lagrange, self.curve))
        writer.write(points_to_bytes(np.stack([self.w, self.u]), self.curve))

    @classmethod
    def read(cls, reader, curve: int) -> "Params":
        """Params::read (commitment.rs:184-205); the 2^(k+1) + 2 square roots run on the device."""
        head = reader.read(4)
        if len(head) != 4:
            raise ValueError("Params.read: truncated header")
        k = int.from_bytes(head, "little")
        if k >= 32:
            raise ValueError("Params.read: k out of range")                 # commitment.rs:41
        n = 1 << k
        raw = reader.read(32 * (2 * n + 2))
        if len(raw) != 32 * (2 * n + 2):
            raise ValueError("Params.read: truncated point data")
        pts = points_from_bytes(raw, curve)
        return cls(curve, k, pts[:n], pts[n:2 * n], pts[2 * n], pts[2 * n + 1])

    def _check_device(self, t) -> None:
        """One process may drive several GPUs: a column must live where this Params' tables were registered (a launch on
        another device would read them through a peer mapping at best)."""
        if t.device.index != self.device_index:
            raise ValueError(f"Params registered on cuda:{self.device_index} used with a tensor on {t.device}")

    def close(self):
        for h in (self._h_g, self._h_gl, self._h_gu, self._h_pair, self._h_guw):
            if h.value:
                lib().h2_bases_free(h)
                h.value = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get_g(self) -> np.ndarray:
        return self.g.copy()

    def _commit(self, handle, poly, r: Blind, affine: bool):
        out_kind = OUT_AFFINE if affine else OUT_JACOBIAN
        out_len = 8 if affine else 12
        if poly.shape[0] != self.n:
            raise ValueError("commit: polynomial length != n")
        if _is_torch(poly):
            import torch
            self._check_device(poly)
            blind = torch.from_numpy(np.ascontiguousarray(r.value).view(np.int64)).to(poly.device)
            out = torch.empty(out_len, dtype=torch.int64, device=poly.device)
            check(lib().h2_commit_device(handle, poly.data_ptr(), self.n, None, blind.data_ptr(),
                                         FORM_MONTGOMERY, out_kind, out.data_ptr(), _stream_ptr()), "h2_commit_device")
            return out
        poly = _np(poly, 4)
        out = np.zeros(out_len, dtype=np.uint64)
        check(lib().h2_commit(handle, _p(poly), self.n, None, _p(np.ascontiguousarray(r.value)), FORM_MONTGOMERY,
                              out_kind, _p(out)), "h2_commit")
        return out

    def commit_batch(self, polys, blinds, lagrange: bool = False, affine: bool = False):
        """The independent column commits of one prover phase (plonk/prover.rs:305-313) in one call: device
        tensors in, one (len, 12|8) device tensor out; columns overlap on internal streams."""
        import torch
        if len(polys) != len(blinds):
            raise ValueError("commit_batch: polys and blinds differ in length")
        if not polys:
            return None
        dev = polys[0].device
        self._check_device(polys[0])
        out_len = 8 if affine else 12
        out = torch.empty((len(polys), out_len), dtype=torch.int64, device=dev)
        d_bl = torch.from_numpy(np.stack([np.ascontiguousarray(b.value) for b in blinds]).view(np.int64)).to(dev)
        n_ = len(polys)
        arr = C.c_void_p * n_
        for p_ in polys:
            if p_.shape[0] != self.n:
                raise ValueError("commit_batch: polynomial length != n")
        sc = arr(*[p_.data_ptr() for p_ in polys])
        bl = arr(*[d_bl[i].data_ptr() for i in range(n_)])
        outs = arr(*[out[i].data_ptr() for i in range(n_)])
        check(lib().h2_commit_batch_device(self._h_gl if lagrange else self._h_g, sc, n_, self.n, None, bl,
                                           FORM_MONTGOMERY, OUT_AFFINE if affine else OUT_JACOBIAN, outs, _stream_ptr()),
              "h2_commit_batch_device")
        return out

    def opening_columns_commit(self, cols, blinds, affine: bool = True):
        """sum_m col[m] * (g || u)[m] + blind * w for each (n + 1)-row CUDA column: the shape of L_j / R_j
        (poly/commitment/prover.rs:107-114) once they are written over the original generators (opening.py)."""
        import torch
        if not self._h_gu.value:
            gu = np.ascontiguousarray(np.concatenate([self.g, self.u.reshape(1, 8)]))
            check(lib().h2_bases_register(self.curve, _p(gu), self.n + 1, FORM_MONTGOMERY, C.byref(self._h_gu)), "h2_bases_register")
            check(lib().h2_bases_set_blind_base(self._h_gu, _p(self.w), FORM_MONTGOMERY), "h2_bases_set_blind_base")
        dev = cols[0].device
        n_ = len(cols)
        out = torch.empty((n_, 8 if affine else 12), dtype=torch.int64, device=dev)
        d_bl = torch.from_numpy(np.stack([np.ascontiguousarray(b, dtype=np.uint64).reshape(4) for b in blinds]).view(np.int64)).to(dev)
        arr = C.c_void_p * n_
        for c_ in cols:
            if c_.shape[0] != self.n + 1 or not c_.is_contiguous():
                raise ValueError("opening_columns_commit: columns must hold n + 1 scalars")
        check(lib().h2_commit_batch_device(self._h_gu, arr(*[c_.data_ptr() for c_ in cols]), n_, self.n + 1, None,
                                           arr(*[d_bl[i].data_ptr() for i in range(n_)]), FORM_MONTGOMERY,
                                           OUT_AFFINE if affine else OUT_JACOBIAN, arr(*[out[i].data_ptr() for i in range(n_)]),
                                           _stream_ptr()), "h2_commit_batch_device")
        return out

    def pair_commit_supported(self) -> bool:
        """Whether L_j and R_j of a round can share one commit (h2_commit_pair_device over g || u || u || w || w): tables from
        8192 points on."""
        return bool(lib().h2_commit_pair_supported(self.n + 4))

    def opening_pair_commit(self, column, pair_shift: int, affine: bool = False):
        """L_j and R_j of an opening-argument round from ONE (n + 4)-row CUDA column over g || u || u || w || w
        (h2_commit_pair_device): rows i < n belong to output (i >> pair_shift) & 1; rows n, n + 1 are the U scalars of L and R,
        rows n + 2, n + 3 their W scalars.  Returns a (2, 12 | 8) CUDA tensor."""
        import torch
        self._opening_basis(True)
        if column.shape[0] != self.n + 4 or not column.is_contiguous():
            raise ValueError("opening_pair_commit: the column must hold n + 4 scalars")
        out = torch.empty((2, 8 if affine else 12), dtype=torch.int64, device=column.device)
        check(lib().h2_commit_pair_device(self._h_pair, column.data_ptr(), self.n + 4, pair_shift, FORM_MONTGOMERY,
                                          OUT_AFFINE if affine else OUT_JACOBIAN, out.data_ptr(), _stream_ptr()), "h2_commit_pair_device")
        return out

    def _opening_basis(self, paired: bool):
        """The registered basis of the opening argument's commits: g || u || u || w || w (one paired commit per round) or
        g || u || w (two commits per round; [value z] U and [rand] W ride in the last two rows of each column)."""
        if paired:
            if not self._h_pair.value:
                tail = np.stack([self.u, self.u, self.w, self.w])
                basis = np.ascontiguousarray(np.concatenate([self.g, tail]))
                check(lib().h2_bases_register(self.curve, _p(basis), self.n + 4, FORM_MONTGOMERY, C.byref(self._h_pair)), "h2_bases_register")
            return self._h_pair
        if not self._h_guw.value:
            guw = np.ascontiguousarray(np.concatenate([self.g, self.u.reshape(1, 8), self.w.reshape(1, 8)]))
            check(lib().h2_bases_register(self.curve, _p(guw), self.n + 2, FORM_MONTGOMERY, C.byref(self._h_guw)), "h2_bases_register")
        return self._h_guw

    def default_hybrid_rounds(self, paired: bool) -> int:
        """After how many rounds the opening argument moves to the collapsed generators (0 = never): the library's choice,
        k - 14 rounds from k = 16 to 20 (a table of 2^14 points is left), 6 at k = 21, 5 beyond (h2_ipa_default_switch_rounds)."""
        return int(lib().h2_ipa_default_switch_rounds(self.k, 1 if paired else 0))

    def opening_rounds(self, d_p, d_b, z, rands, transcript, paired: bool, hybrid_rounds: int | None = None):
        """The round loop of the opening argument (poly/commitment/prover.rs:104-142) through h2_ipa_rounds_device: d_p (p') and
        d_b are (n, 4) CUDA tensors folded in place, `rands` the 2k blinds l_0, r_0, l_1, ... ; the transcript's write_point /
        squeeze_challenge_scalar are called from inside the loop.  Returns (c, f_delta): the final p'[0] and
        sum_j (l_j / u_j + r_j u_j), both (4,) Montgomery limbs.

        hybrid_rounds = J > 0: the first J rounds run over the original generators, then the library reads G'_J off the
        registered table and runs the remaining k - J rounds over it (None: the library's choice)."""
        import torch
        from ._lib import IPA_SWITCH_DEFAULT
        n, k = self.n, self.k
        if d_p.shape[0] != n or d_b.shape[0] != n or not d_p.is_contiguous() or not d_b.is_contiguous():
            raise ValueError("opening_rounds: p' and b must hold n scalars")
        rands = np.ascontiguousarray(rands, dtype=np.uint64).reshape(2 * k, 4)
        z = np.ascontiguousarray(z, dtype=np.uint64).reshape(4)
        dev = d_p.device
        J = IPA_SWITCH_DEFAULT if hybrid_rounds is None else int(hybrid_rounds)
        if J != IPA_SWITCH_DEFAULT and (J < 0 or J >= k or J > 12 or (J and not paired)):
            raise ValueError("opening_rounds: hybrid_rounds must be in [0, min(k - 1, 12)] and needs the paired schedule")
        cb_w, cb_s, user, failure = _transcript_callbacks(transcript)
        col_l = torch.empty((n + (4 if paired else 2), 4), dtype=torch.int64, device=dev)
        col_r = None if paired else torch.empty((n + 2, 4), dtype=torch.int64, device=dev)
        c = np.zeros(4, dtype=np.uint64)
        f = np.zeros(4, dtype=np.uint64)
        uw = np.ascontiguousarray(np.stack([self.u, self.w]), dtype=np.uint64)
        rc = lib().h2_ipa_rounds_device(self.curve, k, J, self._opening_basis(paired), 1 if paired else 0, d_p.data_ptr(), d_b.data_ptr(),
                                        _p(z), _p(rands), _p(uw), col_l.data_ptr(), col_r.data_ptr() if col_r is not None else None,
                                        cb_w, cb_s, user, _p(c), _p(f), _stream_ptr())
        if failure:
            raise failure[0]
        check(rc, "h2_ipa_rounds_device")
        return c, f

    def open(self, p_poly, p_blind: Blind, x_3, s_poly, s_blind: Blind, rands, transcript, paired: bool, hybrid_rounds: int | None = None):
        """`commitment::create_proof` (poly/commitment/prover.rs:26-151) between its rng and its transcript as ONE native call
        (h2_open_device / h2_open_device_host_s / h2_open): the commitment to s_poly, xi and z, P', b, v and the whole round loop.  p_poly, s_poly: both
        (n, 4) CUDA tensors (s_poly is consumed: it becomes p' and is folded in place), both host arrays, or p_poly on the device and s_poly a host array; s_poly, s_blind, rands
        (2k scalars): the randomness in the reference's order.  Returns (c, f), the two scalars the caller writes last (:146-148)."""
        from ._lib import IPA_SWITCH_DEFAULT
        n, k = self.n, self.k
        rands = np.ascontiguousarray(rands, dtype=np.uint64).reshape(2 * k, 4)
        x3 = np.ascontiguousarray(x_3, dtype=np.uint64).reshape(4)
        pb = np.ascontiguousarray(p_blind.value, dtype=np.uint64).reshape(4)
        sb = np.ascontiguousarray(s_blind.value, dtype=np.uint64).reshape(4)
        J = IPA_SWITCH_DEFAULT if hybrid_rounds is None else int(hybrid_rounds)
        if J != IPA_SWITCH_DEFAULT and (J < 0 or J >= k or J > 12 or (J and not paired)):
            raise ValueError("open: hybrid_rounds must be in [0, min(k - 1, 12)] and needs the paired schedule")
        if p_poly.shape[0] != n or s_poly.shape[0] != n:
            raise ValueError("open: p_poly and s_poly must hold n scalars")
        cb_w, cb_s, user, failure = _transcript_callbacks(transcript)
        c = np.zeros(4, dtype=np.uint64)
        f = np.zeros(4, dtype=np.uint64)
        uw = np.ascontiguousarray(np.stack([self.u, self.w]), dtype=np.uint64)
        basis = self._opening_basis(paired)
        if _is_torch(p_poly) and not _is_torch(s_poly):
            # p_poly resident, s_poly from a host rng: its quarters cross PCIe inside the call and are committed as they land
            self._check_device(p_poly)
            if not p_poly.is_contiguous():
                raise ValueError("open: contiguous tensors, please")
            sp = _np(s_poly, 4)
            rc = lib().h2_open_device_host_s(self.curve, k, self._h_g, basis, 1 if paired else 0, J, _p(uw), p_poly.data_ptr(), _p(pb), _p(x3), _p(sp),
                                             _p(sb), _p(rands), cb_w, cb_s, user, _p(c), _p(f), _stream_ptr())
            if failure:
                raise failure[0]
            check(rc, "h2_open_device_host_s")
            return c, f
        if _is_torch(p_poly) != _is_torch(s_poly):
            raise ValueError("open: a host p_poly needs a host s_poly")
        if _is_torch(p_poly):
            self._check_device(p_poly)
            self._check_device(s_poly)
            if not p_poly.is_contiguous() or not s_poly.is_contiguous():
                raise ValueError("open: contiguous tensors, please")
            rc = lib().h2_open_device(self.curve, k, self._h_g, basis, 1 if paired else 0, J, _p(uw), p_poly.data_ptr(), _p(pb), _p(x3),
                                      s_poly.data_ptr(), _p(sb), _p(rands), cb_w, cb_s, user, _p(c), _p(f), _stream_ptr())
            name = "h2_open_device"
        else:
            pp, sp = _np(p_poly, 4), _np(s_poly, 4)
            rc = lib().h2_open(self.curve, k, self._h_g, basis, 1 if paired else 0, J, _p(uw), _p(pp), _p(pb), _p(x3), _p(sp), _p(sb), _p(rands),
                               cb_w, cb_s, user, _p(c), _p(f))
            name = "h2_open"
        if failure:
            raise failure[0]
        check(rc, name)
        return c, f

    def commit_unblinded(self, scalars):
        """sum_i scalars[i] * g[i] with no blind term, Jacobian: the g part of `MSM::eval` (poly/commitment/msm.rs:163-166)."""
        import torch
        if not _is_torch(scalars) or scalars.shape[0] != self.n:
            raise ValueError("commit_unblinded: need n scalars on the device")
        out = torch.empty(12, dtype=torch.int64, device=scalars.device)
        check(lib().h2_commit_device(self._h_g, scalars.data_ptr(), self.n, None, None, FORM_MONTGOMERY, OUT_JACOBIAN, out.data_ptr(),
                                     _stream_ptr()), "h2_commit_device")
        return out

    def commit(self, poly, r: Blind, affine: bool = False):
        """commitment.rs:119-130: sum poly[i] * g[i] + r * w.  `poly`: Polynomial<Coeff> or a raw limb array."""
        from .poly import Coeff, unwrap
        return self._commit(self._h_g, unwrap(poly, Coeff, "Params.commit")[0], r, affine)

    def commit_lagrange(self, poly, r: Blind, affine: bool = False):
        """commitment.rs:135-150: sum poly[i] * g_lagrange[i] + r * w.  `poly`: Polynomial<LagrangeCoeff> or a raw limb array."""
        from .poly import LagrangeCoeff, unwrap
        return self._commit(self._h_gl, unwrap(poly, LagrangeCoeff, "Params.commit_lagrange")[0], r, affine)
