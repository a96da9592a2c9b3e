"""Kernel-level parity: every device operator against the CPU oracle on the
same randomised fields (the device-vs-host differential pattern of the
reference's BiCGStab3D_CL/CL_Matrix.cpp:474-651).  Stencil outputs and
boundary fills are pointwise and must match bit for bit; dot products and
norms differ only by summation order (tolerance 1e-13 relative)."""
import numpy as np
import pytest

from helpers import points
from krylov_amd import capi, problems as P

pytestmark = pytest.mark.gpu

SHAPES = [(16, 16, 16), (20, 12, 9), (40, 33, 17), (130, 20, 18)]
BCS = [(0,) * 6, (1,) * 6, (0, 0, 1, 1, 0, 1), (1, 0, 0, 1, 1, 0), (2, 0, 1, 2, 0, 1)]


def cfg_of(orc, p, **kw):
    return orc.make_cfg(p.mx, p.delx, p.bc, var_coef=p.var_coef, use_offdiag=p.use_offdiag, Dc=p.Dc, **kw)


@pytest.mark.parametrize("mx", SHAPES)
@pytest.mark.parametrize("bc", BCS)
def test_do_bcs_bit_exact(orc, mx, bc):
    p = P.random_fields(mx, seed=1, bc=bc)
    want = p.phi.copy()
    orc.do_bcs(cfg_of(orc, p), want)
    with capi.Solver(p.mx, p.delx, p.bc) as s:
        got = s.op_do_bcs(p.phi.copy())
    assert np.array_equal(got, want)


@pytest.mark.parametrize("mx", SHAPES)
@pytest.mark.parametrize("bc", BCS)
def test_generic_spmv_const_bit_exact(orc, mx, bc):
    p = P.random_fields(mx, seed=2, bc=bc, var_coef=False, Dc=(1.0, 2.5, 10.0))
    psi_o = p.phi.copy()
    want = orc.spmv_const(cfg_of(orc, p), psi_o, p.lam, True)
    with capi.Solver(p.mx, p.delx, p.bc) as s:
        psi_g = p.phi.copy()
        got = s.op_spmv_const(psi_g, p.lam, p.Dc, apply_bcs=True, fused=False)
    assert np.array_equal(psi_g, psi_o)          # the operator's do_BCs on its input
    assert np.array_equal(got, want)


@pytest.mark.parametrize("mx", SHAPES)
@pytest.mark.parametrize("bc", BCS[:3])
@pytest.mark.parametrize("offd", [False, True])
@pytest.mark.parametrize("pow2", [False, True])
def test_generic_spmv_var_bit_exact(orc, mx, bc, offd, pow2):
    delx = (1 / 16, 1 / 32, 1 / 8) if pow2 else None
    p = P.random_fields(mx, seed=3, bc=bc, var_coef=True, use_offdiag=offd, delx=delx)
    want = orc.spmv_var(cfg_of(orc, p), p.phi, p.lam, p.Dxx, p.Dyy, p.Dzz, p.Dxy, True)
    with capi.Solver(p.mx, p.delx, p.bc) as s:
        got = s.op_spmv_var(p.phi, p.lam, p.Dxx, p.Dyy, p.Dzz, p.Dxy, offd, apply_bcs=True, fused=False)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("mx", SHAPES + [(256, 24, 21)])
def test_fused_spmv_const_bit_exact(orc, mx):
    p = P.random_fields(mx, seed=4, var_coef=False, Dc=(1.0, 2.5, 10.0))
    psi_o = p.phi.copy()
    want = orc.spmv_const(cfg_of(orc, p), psi_o, p.lam, True)
    with capi.Solver(p.mx, p.delx, p.bc) as s:
        got = s.op_spmv_const(p.phi.copy(), p.lam, p.Dc, apply_bcs=True, fused=True)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("mx", SHAPES + [(256, 24, 21)])
@pytest.mark.parametrize("offd", [False, True])
@pytest.mark.parametrize("pow2", [False, True])
def test_fused_spmv_var_bit_exact(orc, mx, offd, pow2):
    delx = (1 / 16, 1 / 32, 1 / 8) if pow2 else None
    p = P.random_fields(mx, seed=5, var_coef=True, use_offdiag=offd, delx=delx)
    want = orc.spmv_var(cfg_of(orc, p), p.phi, p.lam, p.Dxx, p.Dyy, p.Dzz, p.Dxy, True)
    with capi.Solver(p.mx, p.delx, p.bc) as s:
        got = s.op_spmv_var(p.phi, p.lam, p.Dxx, p.Dyy, p.Dzz, p.Dxy, offd, apply_bcs=True, fused=True)
    # the fused kernel evaluates interior points; Dirichlet faces are zero in both
    assert np.array_equal(points(got), points(want))


@pytest.mark.parametrize("mx", SHAPES)
def test_dot_and_norm(orc, mx):
    p = P.random_fields(mx, seed=6)
    cfg = cfg_of(orc, p)
    with capi.Solver(p.mx, p.delx, p.bc) as s:
        d = s.op_dot(p.phi, p.rhs)
        n = s.op_l2norm(p.phi)
    dw, nw = orc.dot(cfg, p.phi, p.rhs), orc.l2norm(cfg, p.phi)
    scale = np.abs(points(p.phi) * points(p.rhs)).sum()
    assert abs(d - dw) <= 1e-13 * scale
    assert abs(n - nw) <= 1e-13 * nw


def test_norm_corner_weight_is_quarter(orc):
    """A field that is non-zero only at the eight corners separates the norm's
    corner weight 1/4 (BICGStab3D/Krylov.C:136-143) from the dot's 1/8."""
    p = P.random_fields((8, 8, 8), seed=7)
    v = np.zeros_like(p.phi)
    for k in (1, -2):
        for j in (1, -2):
            for i in (1, -2):
                v[k, j, i] = 3.0
    with capi.Solver(p.mx, p.delx, p.bc) as s:
        assert s.op_l2norm(v) == pytest.approx(np.sqrt(8 * 0.25 * 9.0), rel=1e-15)
        assert s.op_dot(v, v) == pytest.approx(8 * 0.125 * 9.0, rel=1e-15)
