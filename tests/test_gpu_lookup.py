"""The lookup argument's data-dependent step on the device -- `h2_sort_device`, `h2_permute_expression_pair_device`
(plonk/lookup/prover.rs:557-647) -- against the integer restatement of the reference's sequential walk (oracle/lookup.py),
bit-exact; the properties the reference's own sanity check states (:629-642); sizes that are not powers of two and that span
several sort tiles; the ConstraintSystemFailure case.  Runs only on a real MI355X (`-m gpu`)."""
import random

import numpy as np
import pytest
import torch

import halo2_amd as h
from halo2_amd import fields
from halo2_amd._lib import ConstraintSystemFailure
from halo2_amd.arithmetic import permute_expression_pair, sort_field
from oracle import c_oracle as co
from oracle import lookup as olk

pytestmark = pytest.mark.gpu


def _up(ints, field):
    return torch.from_numpy(fields.to_limbs(ints, field, True).view(np.int64)).cuda()


def _dn(t, field):
    return co.limbs_to_ints(co.from_mont(field, t.cpu().numpy().view(np.uint64)))


@pytest.mark.parametrize("field", [h.FP, h.FQ])
@pytest.mark.parametrize("n", [0, 1, 2, 3, 31, 1000, 2048, 2049, 5000, 70001])
def test_sort_matches_python_sorted(field, n):
    m = fields.MODULUS[field]
    rnd = random.Random(n + field)
    vals = [rnd.choice([rnd.randrange(m), rnd.randrange(1 << 20), m - 1 - rnd.randrange(5), 0]) for _ in range(n)]
    t = _up(vals, field) if n else torch.empty((0, 4), dtype=torch.int64, device="cuda")
    sort_field(t, field)
    assert _dn(t, field) == sorted(vals) if n else True


def test_sort_canonical_form_and_large():
    field, n = h.FP, (1 << 20) + 12345
    a = co.random_field(field, 91, n)
    canon = co.from_mont(field, a)
    t = torch.from_numpy(canon.view(np.int64)).cuda()
    from halo2_amd._lib import FORM_CANONICAL
    sort_field(t, field, FORM_CANONICAL)
    host = t.cpu().numpy().view(np.uint64)
    want = canon[np.lexsort((canon[:, 0], canon[:, 1], canon[:, 2], canon[:, 3]))]      # most significant limb last = primary key
    assert np.array_equal(host, want)


def _lookup_case(rnd, m, usable, table_size, dup_heavy):
    table_vals = [rnd.randrange(m) if not dup_heavy else rnd.randrange(50) for _ in range(table_size)]
    table = [rnd.choice(table_vals) for _ in range(usable)]
    for i, v in enumerate(table_vals[:usable]):                      # every table value present at least once (if it fits)
        table[i] = v
    rnd.shuffle(table)
    present = sorted(set(table))
    inputs = [rnd.choice(present) for _ in range(usable)]
    return inputs, table


@pytest.mark.parametrize("field", [h.FP, h.FQ])
@pytest.mark.parametrize("usable,table_size,dup_heavy", [(1, 1, False), (2, 1, False), (26, 4, False), (26, 26, False), (250, 16, True),
                                                          (1018, 256, False), (4090, 100, True), (70000, 1 << 12, False)])
def test_permute_expression_pair_matches_the_sequential_walk(field, usable, table_size, dup_heavy):
    m = fields.MODULUS[field]
    rnd = random.Random(usable * 7 + table_size + field)
    inputs, table = _lookup_case(rnd, m, usable, table_size, dup_heavy)
    extra = [rnd.randrange(m) for _ in range(6)]                      # rows beyond `usable` are ignored
    d_in, d_tb = _up(inputs + extra, field), _up(table + extra[::-1], field)
    a, s = permute_expression_pair(d_in, d_tb, usable, field)
    got_a, got_s = _dn(a, field), _dn(s, field)
    want = olk.permute_expression_pair(inputs, table, usable)
    assert want is not None
    assert got_a == want[0] and got_s == want[1]
    assert _dn(d_in, field) == inputs + extra and _dn(d_tb, field) == table + extra[::-1]        # inputs untouched
    # the reference's own sanity check (:629-642) and the multiset identities the argument relies on
    last = None
    for x_, y_ in zip(got_a, got_s):
        if x_ != y_:
            assert x_ == last
        last = x_
    assert sorted(got_s) == sorted(table) and got_a == sorted(inputs)


def test_missing_table_value_is_a_constraint_system_failure():
    field = h.FP
    inputs, table = [5, 7, 7, 9], [5, 7, 8, 8]
    assert olk.permute_expression_pair(inputs, table, 4) is None
    with pytest.raises(ConstraintSystemFailure):
        permute_expression_pair(_up(inputs, field), _up(table, field), 4, field)
    a, s = permute_expression_pair(_up([5, 7, 7, 8], field), _up(table, field), 4, field)       # and the library is usable afterwards
    assert (_dn(a, field), _dn(s, field)) == olk.permute_expression_pair([5, 7, 7, 8], table, 4)
