"""Host-side pieces added in round 5 that need no GPU: the transcript wrapper that reads queued evaluations back together
(halo2_amd.transcript.DeferredScalars), and the integer identity behind the S commitment by quarters (csrc/ipa.hip: ipa_fix_s0)."""
import random

import numpy as np
import pytest

from halo2_amd.transcript import DeferredScalars, write_evaluation

P = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001


class _Recorder:
    """A transcript that only remembers what it was asked, in order."""
    handle = 77

    def __init__(self):
        self.log = []

    def write_scalar(self, s):
        self.log.append(("scalar", tuple(int(v) for v in np.asarray(s, dtype=np.uint64).reshape(4))))

    def write_point(self, p):
        self.log.append(("point", tuple(int(v) for v in np.asarray(p).reshape(-1))))

    def squeeze_challenge_scalar(self):
        self.log.append(("squeeze",))
        return np.arange(4, dtype=np.uint64)


def test_deferred_scalars_keep_the_order_and_flush_before_anything_else():
    torch = pytest.importorskip("torch")
    inner = _Recorder()
    d = DeferredScalars(inner)
    t1 = torch.tensor([1, 2, 3, 4], dtype=torch.int64)            # a "device" evaluation (a CPU tensor stands in: .cpu() is the read-back)
    t2 = torch.tensor([[5, 6, 7, 8]], dtype=torch.int64)          # (1, 4) is reshaped
    host = np.array([9, 10, 11, 12], dtype=np.uint64)
    write_evaluation(d, t1)
    write_evaluation(d, host)                                     # host scalars keep their place in the queue
    write_evaluation(d, t2)
    assert inner.log == []                                        # nothing has reached the transcript yet
    d.write_point(np.arange(8, dtype=np.uint64))                  # any other use flushes first
    assert [e[0] for e in inner.log] == ["scalar", "scalar", "scalar", "point"]
    assert [e[1] for e in inner.log[:3]] == [(1, 2, 3, 4), (9, 10, 11, 12), (5, 6, 7, 8)]
    write_evaluation(d, t1)
    assert d.handle == 77 and inner.log[-1] == ("scalar", (1, 2, 3, 4))          # asking for the native handle flushes too
    write_evaluation(d, t2)
    d.squeeze_challenge_scalar()
    assert [e[0] for e in inner.log[-2:]] == ["scalar", "squeeze"]
    d.flush()
    d.flush()                                                      # idempotent
    assert len(inner.log) == 7
    # a transcript that does not defer gets the value at once, tensors read back on the spot
    plain = _Recorder()
    write_evaluation(plain, t1)
    write_evaluation(plain, host)
    assert plain.log == [("scalar", (1, 2, 3, 4)), ("scalar", (9, 10, 11, 12))]


def test_deferred_scalars_keep_the_unwritten_tail_when_a_write_fails():
    """Advisor (round 5): a write that raises inside flush() must not drop the scalars behind it, and probing `handle` must surface the error
    instead of reading as "no native handle"."""
    torch = pytest.importorskip("torch")

    class _Failing(_Recorder):
        fail_at = 1

        def write_scalar(self, s):
            if len(self.log) == self.fail_at:
                self.fail_at = -1
                raise RuntimeError("transcript write failed")
            super().write_scalar(s)

    inner = _Failing()
    d = DeferredScalars(inner)
    for v in (1, 2, 3):
        write_evaluation(d, torch.tensor([v, 0, 0, 0], dtype=torch.int64))
    with pytest.raises(RuntimeError):
        d.handle                                                   # the probe flushes: the failure is reported, not turned into a default
    assert [e[1][0] for e in inner.log] == [1] and len(d.queue) == 2           # the failing scalar and the one behind it are still queued
    write_evaluation(d, np.array([4, 0, 0, 0], dtype=np.uint64))
    d.flush()                                                      # the retry writes the missing tail, in order
    assert [e[1][0] for e in inner.log] == [1, 2, 3, 4] and d.queue == []


def test_evaluation_from_the_quarters():
    """ipa_fix_s0: s(x) = sum_r x^(r n / 4) ev_r with ev_r the evaluation of quarter r as a polynomial in its own index, and x^(n / 4) by k - 2 squarings;
    after s[0] -= s(x) the polynomial has its root at x (prover.rs:49-51)."""
    rng = random.Random(11)
    for k in (4, 7, 10):
        n, q = 1 << k, 1 << (k - 2)
        s = [rng.randrange(P) for _ in range(n)]
        x = rng.randrange(P)
        ev = [sum(c * pow(x, i, P) for i, c in enumerate(s[r * q:(r + 1) * q])) % P for r in range(4)]
        xq = x
        for _ in range(k - 2):
            xq = xq * xq % P
        assert xq == pow(x, q, P)
        total = (ev[0] + xq * ev[1] + xq * xq % P * ev[2] + pow(xq, 3, P) * ev[3]) % P
        assert total == sum(c * pow(x, i, P) for i, c in enumerate(s)) % P
        s[0] = (s[0] - total) % P
        assert sum(c * pow(x, i, P) for i, c in enumerate(s)) % P == 0
