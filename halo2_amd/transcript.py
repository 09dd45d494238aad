"""Host-side mirror of the reference's Fiat-Shamir transcript (halo2_proofs/src/transcript.rs:150-300):
`Blake2bWrite` / `Blake2bRead` with `Challenge255`.  Pure host logic (hashing a few hundred bytes per proof);
it exists so the device-resident opening argument (halo2_amd/opening.py) can be driven exactly as the reference
drives `create_proof`, and so tests read like the reference's `test_opening_proof` (poly/commitment.rs:305-379).

Points are affine (8,) and scalars (4,) uint64 Montgomery limbs, as everywhere in this package."""
from __future__ import annotations

import hashlib

import numpy as np

from . import fields

_PREFIX_CHALLENGE, _PREFIX_POINT, _PREFIX_SCALAR = b"\x00", b"\x01", b"\x02"     # transcript.rs:14-20


def _le32(v: int) -> bytes:
    return int(v).to_bytes(32, "little")


def point_to_bytes(x: int, y: int) -> bytes:
    """pasta_curves `to_bytes` (compressed): x little-endian, top bit = parity of y."""
    b = bytearray(_le32(x))
    b[31] |= (y & 1) << 7
    return bytes(b)


class _Blake2bTranscript:
    def __init__(self, curve: int):
        self.curve = curve
        self.base, self.scalar = fields.CURVE_FIELDS[curve]
        self.state = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")    # transcript.rs:162-166

    # -- Transcript -------------------------------------------------------------------------------
    def squeeze_challenge(self) -> int:
        """transcript.rs:200-205 + Challenge255::new (:286-296): 64 hash bytes reduced into the scalar field."""
        self.state.update(_PREFIX_CHALLENGE)
        digest = self.state.copy().digest()
        return int.from_bytes(digest, "little") % fields.MODULUS[self.scalar]

    def squeeze_challenge_scalar(self) -> np.ndarray:
        return fields.scalar_limbs(self.squeeze_challenge(), self.scalar, True)

    def _coords(self, point) -> tuple[int, int]:
        """Affine (8 limbs) or Jacobian (12 limbs, what `C::Curve` is) -> canonical affine integers.  The Jacobian case is
        the prover's `.to_affine()` before `write_point` (poly/commitment/prover.rs:116-117): one modular inversion on the
        host costs microseconds, on the device a 255-step dependent chain on a single lane."""
        point = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1)
        if point.shape[0] == 12:
            xj, yj, zj = fields.from_limbs(point.reshape(3, 4), self.base, True)
            m = fields.MODULUS[self.base]
            if zj == 0:
                raise ValueError("cannot write points at infinity to the transcript")    # transcript.rs:209-214
            zi = pow(zj, -1, m)
            zi2 = zi * zi % m
            return xj * zi2 % m, yj * zi2 * zi % m
        x, y = fields.from_limbs(point.reshape(2, 4), self.base, True)
        if x == 0 and y == 0:
            raise ValueError("cannot write points at infinity to the transcript")        # transcript.rs:209-214
        return x, y

    def common_point(self, point):
        x, y = self._coords(point)
        self.state.update(_PREFIX_POINT)
        self.state.update(_le32(x))
        self.state.update(_le32(y))

    def common_scalar(self, scalar):
        self.state.update(_PREFIX_SCALAR)
        self.state.update(_le32(fields.from_limbs(scalar, self.scalar, True)[0]))


class Blake2bWrite(_Blake2bTranscript):
    def __init__(self, curve: int):
        super().__init__(curve)
        self.writer = bytearray()

    def write_point(self, point):
        self.common_point(point)
        self.writer += point_to_bytes(*self._coords(point))                          # transcript.rs:183-187

    def write_scalar(self, scalar):
        self.common_scalar(scalar)
        self.writer += _le32(fields.from_limbs(scalar, self.scalar, True)[0])        # transcript.rs:188-192

    def finalize(self) -> bytes:
        return bytes(self.writer)
