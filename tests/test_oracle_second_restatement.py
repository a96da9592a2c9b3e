"""The C oracle against a second restatement of the reference, written in numpy from the
reference's source text alone (oracle/restatement_np.py).  Both evaluate the reference's
expressions with the same roundings and sum in the same order, so every number must agree bit
for bit: operators, boundary treatment, reductions, and whole solves (history, pass count,
the complete returned array including ghost cells)."""
import numpy as np
import pytest

from helpers import oracle_solve
from krylov_amd import problems as P
from oracle import restatement_np as R

BCS = [(0,) * 6, (1,) * 6, (0, 0, 1, 1, 0, 1), (1, 0, 0, 1, 1, 0), (0, 1, 1, 0, 1, 1)]
SHAPES = [(8, 8, 8), (12, 10, 14), (5, 9, 7)]


def fields(mx, seed):
    p = P.random_fields(mx, seed=seed)
    return p


@pytest.mark.parametrize("bc", BCS)
@pytest.mark.parametrize("mx", SHAPES)
def test_boundary_treatment(orc, mx, bc):
    p = fields(mx, 1)
    g = R.Grid(p.mx, p.delx, bc)
    cfg = orc.make_cfg(p.mx, p.delx, bc)
    a, b = p.rhs.copy(), p.rhs.copy()
    orc.do_bcs(cfg, a)
    R.do_bcs(g, b)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("mx", SHAPES)
def test_reductions(orc, mx):
    p = fields(mx, 2)
    g = R.Grid(p.mx, p.delx, (0,) * 6)
    cfg = orc.make_cfg(p.mx, p.delx)
    assert orc.dot(cfg, p.rhs, p.lam) == R.dot(g, p.rhs, p.lam)
    assert orc.l2norm(cfg, p.rhs) == R.l2norm(g, p.rhs)
    assert R.l2norm(g, np.zeros_like(p.rhs)) == 0. == orc.l2norm(cfg, np.zeros_like(p.rhs))


@pytest.mark.parametrize("bc", BCS[:3])
@pytest.mark.parametrize("offd", [False, True])
@pytest.mark.parametrize("mx", SHAPES)
def test_operators(orc, mx, offd, bc):
    p = fields(mx, 3)
    g = R.Grid(p.mx, p.delx, bc, Dc=(1., 2., 3.))
    dxy = p.Dxy if offd else np.zeros_like(p.Dxy)
    cfg = orc.make_cfg(p.mx, p.delx, bc, var_coef=True, use_offdiag=offd)
    want = orc.spmv_var(cfg, p.phi.copy(), p.lam, p.Dxx, p.Dyy, p.Dzz, dxy, True)
    got = np.zeros_like(p.phi)
    R.multiply_var(g, p.phi.copy(), p.lam, p.Dxx, p.Dyy, p.Dzz, dxy, got, True)
    pts = (slice(1, -1),) * 3
    assert np.array_equal(want[pts], got[pts])
    cfg = orc.make_cfg(p.mx, p.delx, bc, Dc=(1., 2., 3.))
    psi_a, psi_b = p.phi.copy(), p.phi.copy()
    want = orc.spmv_const(cfg, psi_a, p.lam, True)
    got = np.zeros_like(p.phi)
    R.multiply_const(g, psi_b, p.lam, got, True)
    assert np.array_equal(want[pts], got[pts])
    assert np.array_equal(psi_a, psi_b)            # the input went through do_BCs on both sides


def problem(kind, mx, bc, seed):
    rng = np.random.default_rng(seed)
    if kind == "const":
        p = P.reference_test1(mx)
        p.bc = bc
    elif kind == "var":
        p = P.smooth_var(mx, bc=bc)
    else:
        p = P.offdiag_var(mx, bc=bc)
        p.Dxy = 0.05 + 0.1 * rng.uniform(0, 1, p.Dxy.shape)       # away from zero on the faces
    p.rhs = rng.uniform(-1, 1, p.rhs.shape)
    p.phi = rng.uniform(-1, 1, p.phi.shape)                       # values on faces and ghost cells too
    return p


@pytest.mark.parametrize("L", [1, 2, 3])
@pytest.mark.parametrize("bc", BCS)
@pytest.mark.parametrize("kind", ["const", "var", "offdiag"])
def test_whole_solves(orc, kind, bc, L):
    p = problem(kind, (10, 8, 12), bc, 5)
    want = oracle_solve(orc, p, L=L, eps=1e-10, max_outer=10)
    phi = p.phi.copy()
    g = R.Grid(p.mx, p.delx, bc, Dc=p.Dc)
    got = R.solve(g, L, 1e-10, phi, p.rhs.copy(), p.lam.copy(),
                  *( (p.Dxx.copy(), p.Dyy.copy(), p.Dzz.copy(), p.Dxy.copy()) if p.var_coef else () ),
                  use_offdiag=p.use_offdiag, max_outer=10)
    assert got["norm_rhs"] == want["norm_rhs"]
    assert got["norm_init"] == want["norm_init"]
    assert got["iters"] == want["iters"] >= 3
    assert np.array_equal(np.array(got["history"]), want["history"])
    assert np.array_equal(phi, want["phi"])


def test_converged_solve_and_zero_rhs(orc):
    p = P.reference_test1(12)
    want = oracle_solve(orc, p, L=2, eps=1e-8)
    phi = p.phi.copy()
    got = R.solve(R.Grid(p.mx, p.delx, p.bc, Dc=p.Dc), 2, 1e-8, phi, p.rhs.copy(), p.lam.copy())
    assert got["status"] == want["status"] == 0 and got["iters"] == want["iters"] > 3
    assert np.array_equal(np.array(got["history"]), want["history"]) and np.array_equal(phi, want["phi"])
    p.rhs[:] = 0
    p.phi[:] = 0
    got = R.solve(R.Grid(p.mx, p.delx, p.bc, Dc=p.Dc), 2, 1e-8, p.phi.copy(), p.rhs.copy(), p.lam.copy())
    want = oracle_solve(orc, p, L=2, eps=1e-8)
    assert got["iters"] == want["iters"] == 0 and got["norm_rhs"] == want["norm_rhs"] == 1.
