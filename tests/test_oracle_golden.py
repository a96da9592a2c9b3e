"""The CPU oracle against the committed fixtures (tests/golden/oracle_cases.json),
bit for bit.  The fixtures pin the oracle itself: the reference ships no golden
vectors for this path."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import gen_golden  # noqa: E402

with open(os.path.join(HERE, "golden", "oracle_cases.json")) as f:
    GOLD = json.load(f)["cases"]


def test_generator_and_fixture_agree_on_case_list():
    assert [c[0] for c in gen_golden.CASES] == [c["name"] for c in GOLD]


@pytest.mark.parametrize("case", gen_golden.CASES, ids=[c[0] for c in gen_golden.CASES])
def test_oracle_reproduces_fixture(orc, case):
    got = gen_golden.run_case(*case)
    want = next(c for c in GOLD if c["name"] == case[0])
    assert got["status"] == want["status"]
    assert got["iters"] == want["iters"]
    for key in ("norm_rhs", "norm_init", "phi_sum", "phi_abs_sum"):
        assert got[key] == want[key], key
    assert got["history"] == want["history"]
    assert got["phi_centre_line"] == want["phi_centre_line"]


def test_hash_field_is_exact():
    a = gen_golden.hash_field((5, 4, 3), 7)
    assert a.min() >= -1.0 and a.max() < 1.0
    assert np.all(a * (1 << 20) == np.round(a * (1 << 20)))
