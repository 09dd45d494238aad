"""The paired commit over small tables (csrc/msm.hip: pair_subdigit_launch) as integer arithmetic on the CPU: the cut of
a scalar into signed 16-bit table digits and of every digit into two signed 8-bit sub-digits (for_each_subdigit), the slot layout the sort
writes its boundaries in (sub_scan: 4 slices x 129 slots, one gap slot per slice), and the bit planes sub_planes sums (plane t = the buckets
with bit t of b + 1 set, doubled t times; plane 7 = bucket 127; position 1 doubled eight more times).  Each function below restates the
kernel's lines; the assertions are the identities the kernels rely on."""
import random

Q = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001


def for_each_subdigit(s, side):
    """(key, window, negative) triples, as the device function emits them for a canonical scalar s < 2^255."""
    out, carry = [], 0
    for w in range(16):
        raw = ((s >> (16 * w)) & 0xFFFF) + carry
        neg = raw > 0x8000
        carry = 1 if neg else 0
        mag = 0x10000 - raw if neg else raw
        e0, c8, e0neg = mag & 255, 0, False
        if e0 > 128:
            e0, e0neg, c8 = 256 - e0, True, 1
        e1 = (mag >> 8) + c8
        assert 0 <= e0 <= 128 and 0 <= e1 <= 128
        if e0:
            out.append((side * 256 + e0 - 1, w, neg != e0neg))
        if e1:
            out.append((side * 256 + 128 + e1 - 1, w, neg))
    assert carry == 0                                    # below 2^255 the top window takes the last carry
    return out


def value_of(entries, side):
    """What the buckets, the planes and the join make of the entries of one side: key = side * 256 + pos * 128 + b weighs (b + 1) * 256^pos,
    the point is row `window` of the table, 2^(16 window) * P."""
    total = 0
    for key, w, negative in entries:
        assert key // 256 == side
        pos, b = (key >> 7) & 1, key & 127
        v = (b + 1) * (256 ** pos) << (16 * w)
        total += -v if negative else v
    return total


def test_sub_digits_add_up_to_the_scalar():
    rng = random.Random(0x5B)
    windows = [0x0000, 0x0001, 0x007F, 0x0080, 0x0081, 0x00FF, 0x0100, 0x0101, 0x7F80, 0x7F81, 0x7FFF, 0x8000, 0x8001, 0x80FF, 0xFF7F, 0xFF80, 0xFF81, 0xFFFF]
    cases = [0, 1, Q - 1, Q - 2, (1 << 255) - 1, 1 << 254]
    for v in windows:
        cases += [(v << (16 * p)) % (1 << 255) for p in range(16)] + [sum(v << (16 * p) for p in range(16)) % (1 << 255)]
    cases += [rng.randrange(Q) for _ in range(5000)]
    most = 0
    for s in cases:
        for side in (0, 1):
            e = for_each_subdigit(s, side)
            assert value_of(e, side) == s, hex(s)
            assert all(side * 256 <= k < side * 256 + 256 for k, _, _ in e)
            most = max(most, len(e))
    assert most <= 32                                     # two entries per digit: what the staging of the sort is sized for


def test_slot_layout_and_boundaries_of_the_sort():
    """sub_scan, a lane per key: slot = key + key / 128; starts[slot] = the key's exclusive prefix; a slice's gap slot starts where the next
    slice starts (an empty bucket); starts[516] = the total, starts[517] = the sentinel msm_accumulate reads past the last boundary."""
    rng = random.Random(3)
    counts = [rng.randrange(0, 50) for _ in range(512)]
    counts[17] = counts[300] = 0
    incl, run = [], 0
    for c in counts:
        run += c
        incl.append(run)
    starts = [None] * 518
    for k in range(512):
        slot = k + (k >> 7)
        starts[slot] = incl[k] - counts[k]
        if k & 127 == 127:
            starts[slot + 1] = incl[k]
        if k == 511:
            starts[516], starts[517] = incl[k], 0xFFFFFFFF
    assert None not in starts and starts[0] == 0 and starts[516] == sum(counts)
    assert all(starts[i] <= starts[i + 1] for i in range(517))
    for k in range(512):                                   # every key's run is exactly its slot's interval; gap slots are empty
        slot = k + (k >> 7)
        assert starts[slot + 1] - starts[slot] == counts[k]
    for y in range(4):
        assert starts[129 * y + 129] - starts[129 * y + 128] == 0 if y < 3 else starts[516] - starts[515] == 0
        assert 129 * y + 128 not in {k + (k >> 7) for k in range(512)}


def test_bit_planes_weigh_bucket_b_by_b_plus_one():
    """sub_planes: plane t < 7 gathers index(k) = (((k >> t) << (t + 1)) | (1 << t) | (k & ((1 << t) - 1))) - 1 for k < 64 and is doubled t times;
    plane 7 is bucket 127 doubled seven times: every bucket b < 128 ends up with the weight b + 1, and a position-1 slice with 256 (b + 1)."""
    weight = [0] * 128
    for t in range(7):
        seen = set()
        for k in range(64):
            b = (((k >> t) << (t + 1)) | (1 << t) | (k & ((1 << t) - 1))) - 1
            assert 0 <= b < 127 and b not in seen and ((b + 1) >> t) & 1
            seen.add(b)
            weight[b] += 1 << t
        assert len(seen) == 64
    weight[127] += 1 << 7
    assert weight == [b + 1 for b in range(128)]
    assert [w << 8 for w in weight] == [256 * (b + 1) for b in range(128)]


def _halves(k, curve):
    """glv.cuh's split of a canonical scalar, with the constants tests/test_glv_constants.py holds the kernel to."""
    import test_glv_constants as t
    base, q, fs_is_fq = (t.FP, t.FQ, True) if curve == "pallas" else (t.FQ, t.FP, False)
    a1, b1m, a2, b2 = (t._array(n, fs_is_fq) for n in ("a1", "b1", "a2", "b2"))
    g1, g2 = t._array("g1", fs_is_fq), t._array("g2", fs_is_fq)
    c1, c2 = (k * g1) >> 256, (k * g2) >> 256
    k1, k2 = k - c1 * a1 - c2 * a2, c1 * b1m - c2 * b2
    lam = a1 * pow(b1m, -1, q) % q
    assert (k1 + k2 * lam - k) % q == 0
    return k1, k2, lam, q


def for_each_subdigit_glv(k1, k2, side):
    """(key, ROW, negative) triples over an endomorphism table (rows 0 .. 8: 2^(16 w) P, rows 9 .. 17: their images under phi), as the device
    function emits them: windows 0 .. 6 of each half signed, window 7 unsigned with its high sub-digit cut evenly into pieces of at most 128,
    window 8 (bit 128 on) unsigned."""
    out = []
    for part, half in enumerate((k1, k2)):
        mag, hn, carry = abs(half), half < 0, 0
        for w in range(7):
            raw = ((mag >> (16 * w)) & 0xFFFF) + carry
            neg = raw > 0x8000
            carry = 1 if neg else 0
            m = 0x10000 - raw if neg else raw
            e0, c8, e0neg = m & 255, 0, False
            if e0 > 128:
                e0, e0neg, c8 = 256 - e0, True, 1
            e1 = (m >> 8) + c8
            dn = neg != hn
            if e0:
                out.append((side * 256 + e0 - 1, part * 9 + w, dn != e0neg))
            if e1:
                out.append((side * 256 + 128 + e1 - 1, part * 9 + w, dn))
        raw = ((mag >> 112) & 0xFFFF) + carry
        e0, c8, e0neg = raw & 255, 0, False
        if e0 > 128:
            e0, e0neg, c8 = 256 - e0, True, 1
        e1 = (raw >> 8) + c8
        if e0:
            out.append((side * 256 + e0 - 1, part * 9 + 7, hn != e0neg))
        pieces = (e1 + 127) // 128
        for j in range(pieces):
            d = (e1 + j) // pieces
            assert 1 <= d <= 128
            out.append((side * 256 + 128 + d - 1, part * 9 + 7, hn))
        rest = min(mag >> 128, 256)
        while rest:
            d = min(rest, 128)
            rest -= d
            out.append((side * 256 + d - 1, part * 9 + 8, hn))
    return out


def test_sub_digits_over_the_endomorphism_table_add_up():
    """sum of (+-)(b + 1) 256^pos 2^(16 (row mod 9)) lambda^(row div 9) over the entries = the scalar, for both curves; at most 40 entries per scalar
    (what the entry list is sized for); no bucket far above the average (the two cuts that WOULD pile entries up -- a carry window, "128 and the
    rest" -- are the ones the kernel avoids: profiles/r05_glv_table.txt)."""
    import collections
    rng = random.Random(0x61)
    for curve in ("pallas", "vesta"):
        hist = collections.Counter()
        n_scalars = 3000
        _, _, lam, q = _halves(1, curve)
        cases = [0, 1, q - 1, lam, q - lam, (q - 1) // 2] + [rng.randrange(q) for _ in range(n_scalars)]
        for k in cases:
            k1, k2, lam, q = _halves(k, curve)
            assert abs(k1) < 1 << 129 and abs(k2) < 1 << 129
            e = for_each_subdigit_glv(k1, k2, 1)
            assert len(e) <= 40
            total = 0
            for key, row, negative in e:
                assert key // 256 == 1 and 0 <= row < 18
                pos, b = (key >> 7) & 1, key & 127
                v = (b + 1) * (256 ** pos) * (1 << (16 * (row % 9))) * (lam if row >= 9 else 1)
                total += -v if negative else v
                hist[key & 255] += 1
            assert (total - k) % q == 0, hex(k)
        avg = sum(hist.values()) / 256
        assert max(hist.values()) < 1.6 * avg, (curve, max(hist.values()), avg)
