"""Host logic of bench.py: slab fields agree with the undivided fields, the
per-kernel byte model adds up to no more than the contract figure."""
import numpy as np
import pytest

import bench


@pytest.mark.parametrize("workload", ["var", "const"])
@pytest.mark.parametrize("nslabs", [2, 4])
def test_slab_fields_tile_the_global_fields(workload, nslabs):
    n = 16
    whole = bench.slab_problem(n, 1, 0, workload)
    mzl = n // nslabs
    for s in range(nslabs):
        part = bench.slab_problem(n, nslabs, s, workload)
        assert part["mx"] == (n, n, mzl)
        for key in ("rhs", "lam") + (("Dxx", "Dyy", "Dzz") if workload == "var" else ()):
            # local plane k (array index k+1) is global plane s*mzl + k, ghosts included
            assert np.array_equal(part[key], whole[key][s * mzl: s * mzl + mzl + 3]), key


def test_kernel_byte_model_totals():
    # doubles per point and pass under the fully fused schedule: 20/26 (l=1), 50/62 (l=2)
    assert sum(bench.kernel_words(1, False, False).values()) == 19      # contract: 20
    assert sum(bench.kernel_words(1, True, False).values()) == 25       # contract: 26
    assert sum(bench.kernel_words(2, False, False).values()) <= 50
    assert sum(bench.kernel_words(2, True, False).values()) <= 62
