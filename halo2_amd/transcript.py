"""Host-side mirror of the reference's Fiat-Shamir transcript (halo2_proofs/src/transcript.rs:150-300):
`Blake2bWrite` / `Blake2bRead` with `Challenge255`.  Pure host logic (hashing a few hundred bytes per proof), kept in the
library (h2_transcript_*) so that the opening argument's native round loop reaches it without a Python callback;
this module is the reference-shaped front: tests read like `test_opening_proof` (poly/commitment.rs:305-379).

Points are affine (8,) and scalars (4,) uint64 Montgomery limbs, as everywhere in this package."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import fields
from ._lib import H2_ERR_ARGS, check, lib, u64p

_PREFIX_CHALLENGE, _PREFIX_POINT, _PREFIX_SCALAR = b"\x00", b"\x01", b"\x02"     # transcript.rs:14-20


def _le32(v: int) -> bytes:
    return int(v).to_bytes(32, "little")


def point_to_bytes(x: int, y: int) -> bytes:
    """pasta_curves `to_bytes` (compressed): x little-endian, top bit = parity of y."""
    b = bytearray(_le32(x))
    b[31] |= (y & 1) << 7
    return bytes(b)


class _Blake2bTranscript:
    """State and hashing live in the library (h2_transcript_*: BLAKE2b-512, `Halo2-Transcript`), so the opening argument's
    round loop can absorb L_j / R_j and squeeze its challenges without coming back to Python."""

    def __init__(self, curve: int):
        self.curve = curve
        self.base, self.scalar = fields.CURVE_FIELDS[curve]
        self._h = C.c_uint64(0)
        check(lib().h2_transcript_new(curve, C.byref(self._h)), "h2_transcript_new")                 # transcript.rs:162-166

    def __del__(self):
        try:
            if self._h.value:
                lib().h2_transcript_free(self._h)
                self._h.value = 0
        except Exception:
            pass

    @property
    def handle(self) -> int:
        return int(self._h.value)

    # -- Transcript -------------------------------------------------------------------------------
    def squeeze_challenge_scalar(self) -> np.ndarray:
        """transcript.rs:200-205 + Challenge255::new (:286-296): 64 hash bytes reduced into the scalar field."""
        out = np.zeros(4, dtype=np.uint64)
        check(lib().h2_transcript_squeeze_challenge(self._h, out.ctypes.data_as(u64p)), "h2_transcript_squeeze_challenge")
        return out

    def squeeze_challenge(self) -> int:
        return fields.from_limbs(self.squeeze_challenge_scalar(), self.scalar, True)[0]

    def _absorb_point(self, point, write: bool):
        """Affine (8 limbs) or Jacobian (12 limbs, what `C::Curve` is: the prover's `.to_affine()` before `write_point`,
        poly/commitment/prover.rs:116-117)."""
        point = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1)
        if point.shape[0] not in (8, 12):
            raise ValueError("transcript: a point is 8 (affine) or 12 (Jacobian) limbs")
        fn = lib().h2_transcript_write_point if write else lib().h2_transcript_common_point
        rc = fn(self._h, point.ctypes.data_as(u64p), 1 if point.shape[0] == 12 else 0)
        if rc == H2_ERR_ARGS:
            raise ValueError("cannot write points at infinity to the transcript")            # transcript.rs:209-214
        check(rc, "h2_transcript_write_point")

    def common_point(self, point):
        self._absorb_point(point, False)

    def common_scalar(self, scalar):
        s = np.ascontiguousarray(scalar, dtype=np.uint64).reshape(4)
        check(lib().h2_transcript_common_scalar(self._h, s.ctypes.data_as(u64p)), "h2_transcript_common_scalar")


class Blake2bWrite(_Blake2bTranscript):
    def write_point(self, point):
        self._absorb_point(point, True)                                                  # transcript.rs:183-187

    def write_scalar(self, scalar):
        s = np.ascontiguousarray(scalar, dtype=np.uint64).reshape(4)
        check(lib().h2_transcript_write_scalar(self._h, s.ctypes.data_as(u64p)), "h2_transcript_write_scalar")   # :188-192

    def finalize(self) -> bytes:
        n = C.c_size_t(0)
        check(lib().h2_transcript_bytes(self._h, None, 0, C.byref(n)), "h2_transcript_bytes")
        buf = (C.c_uint8 * max(n.value, 1))()
        check(lib().h2_transcript_bytes(self._h, buf, n.value, C.byref(n)), "h2_transcript_bytes")
        return bytes(buf[:n.value])


import os as _os
_DEFER_OFF = _os.environ.get("H2_PLONK_DEFER", "1") == "0"


class DeferredScalars:
    """A transcript seen through a queue of scalars that are still in HBM.  The prover writes runs of evaluations to the transcript with no
    challenge between them (plonk/prover.rs:602-675: instance, advice, fixed, vanishing, permutation and lookup evaluations; multiopen/prover.rs:108-110);
    read back one by one, every evaluation is a launch, a 32-byte copy and a stream synchronisation -- 27 of them in the simple-example proof.
    Through this wrapper `write_scalar` accepts the (4,) device tensor itself and queues it; everything queued crosses in ONE copy at `flush()`,
    which any other use of the transcript (a point, a challenge, its handle, its bytes) triggers first -- the bytes written are the same, in the
    same order.  Host scalars may be queued too (they keep their place)."""
    defers = True

    def __init__(self, inner):
        self.inner, self.queue = inner, []

    def write_scalar(self, scalar) -> None:
        self.queue.append(scalar)
        if _DEFER_OFF:                              # H2_PLONK_DEFER=0: every scalar is read back where it is written (the A/B arm)
            self.flush()

    def flush(self) -> None:
        if not self.queue:
            return
        import torch
        dev = [t.reshape(4) for t in self.queue if isinstance(t, torch.Tensor)]
        landed = iter(torch.stack(dev).cpu().numpy().view(np.uint64)) if dev else iter(())      # (a failed read-back leaves the queue as it was)
        host = [next(landed) if isinstance(t, torch.Tensor) else t for t in self.queue]
        self.queue = []
        for i, v in enumerate(host):
            try:
                self.inner.write_scalar(v)
            except BaseException:
                # what was not written stays queued -- as host values, in order, in front of anything queued since: the transcript is never
                # left silently truncated, and a retry (or the caller's error path) sees exactly the missing tail
                self.queue = host[i:] + self.queue
                raise

    @property
    def handle(self):                              # an explicit probe: a failure inside flush() must not look like "no native handle"
        self.flush()
        return getattr(self.inner, "handle", 0)

    def __getattr__(self, name):                   # whatever else the transcript offers, after what is queued
        if name in ("inner", "queue"):             # (not yet constructed: never recurse through flush)
            raise AttributeError(name)
        self.flush()
        return getattr(self.inner, name)


def write_evaluation(transcript, value) -> None:
    """`transcript.write_scalar(eval)` for an evaluation that is a (4,) device tensor: queued when the transcript defers (DeferredScalars),
    read back now otherwise."""
    if getattr(transcript, "defers", False) or isinstance(value, np.ndarray):
        transcript.write_scalar(value)
    else:
        transcript.write_scalar(value.cpu().numpy().view(np.uint64))
