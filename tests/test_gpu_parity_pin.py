"""The solution criterion against the oracle in the REFERENCE's summation order.

Residual histories of BiCGStab(l) separate chaotically under any re-ordering of the
dot products (test_gpu_bitwise.py::test_summation_order_alone_moves_the_history), but
both runs converge to the solution of the same linear system: solved to a tight
tolerance, the device's solution and the reference-order oracle's agree to 1e-9 in the
relative L2 norm over the grid points.  The first pass is also compared (before any
amplification), and the stopping criterion (solveLin_BICGStab.C:496)."""
import numpy as np
import pytest

from helpers import oracle_solve, points, rel_l2
from krylov_amd import capi, problems as P

pytestmark = pytest.mark.gpu

SOLUTION_TOL = 1e-9      # relative L2, device vs reference-order oracle
EPS = 1e-12

CASES = [("ref_test1", 64, 1), ("ref_test1", 64, 2), ("poisson", 64, 1), ("smooth_var", 32, 1), ("smooth_var", 32, 2),
         ("ref_test2", 32, 2), ("ref_test3", 32, 2)]


def make(kind, n):
    return {"ref_test1": P.reference_test1, "poisson": P.poisson, "smooth_var": P.smooth_var,
            "ref_test2": P.reference_test2, "ref_test3": P.reference_test3}[kind](n)


@pytest.mark.parametrize("kind,n,L", CASES)
def test_solution_agrees_with_the_reference_order_oracle(orc, kind, n, L):
    p = make(kind, n)
    want = oracle_solve(orc, p, L=L, eps=EPS)
    got = capi.solve_problem(p, L=L, eps=EPS)
    assert got["fused"]
    assert want["status"] == 0 and got["status"] == 0
    assert got["history"][-1] <= EPS * want["norm_rhs"] and want["history"][-1] <= EPS * want["norm_rhs"]
    first_tol = 1e-8 if (kind == "ref_test1" and L == 2) else 1e-10   # (that rhs is almost an eigenvector)
    assert abs(got["history"][0] - want["history"][0]) <= first_tol * want["norm_rhs"]
    assert rel_l2(points(got["phi"]), points(want["phi"])) <= SOLUTION_TOL


def test_solution_agrees_on_mixed_faces(orc):
    p = P.smooth_var(32, bc=(0, 0, 1, 1, 0, 1))
    want = oracle_solve(orc, p, L=2, eps=EPS)
    got = capi.solve_problem(p, L=2, eps=EPS)
    assert got["path"] == 2 and want["status"] == 0 and got["status"] == 0
    assert rel_l2(points(got["phi"]), points(want["phi"])) <= SOLUTION_TOL
