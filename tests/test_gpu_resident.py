"""Repeated solves with the operator resident on the device (kry_update_* / kry_reset /
kry_resolve, and the adaptor class's set_KeepCoefficients): the bits of full solves, without
re-staging the coefficient fields."""
import ctypes as C
import os

import numpy as np
import pytest

from krylov_amd import capi, problems as P

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


@pytest.mark.parametrize("kind", ["var", "const", "mixed", "slabs"])
def test_ten_time_steps_equal_ten_full_solves(kind):
    if kind == "const":
        p = P.reference_test1(32)
    elif kind == "mixed":
        p = P.smooth_var(24, bc=(0, 0, 1, 1, 0, 1))
    else:
        p = P.smooth_var(32)
    nslabs = 2 if kind == "slabs" else 1
    rng = np.random.default_rng(4)
    steps = [(p.rhs * (1.0 + 0.3 * t), 1e-3 * t * rng.uniform(-1, 1, p.phi.shape)) for t in range(10)]

    def full(rhs, phi0):
        with capi.Solver(p.mx, p.delx, p.bc, L=2, eps=1e-8, nslabs=nslabs, slab_count=nslabs, devices=(0,)) as s:
            phi = phi0.copy()
            if p.var_coef:
                s.solve_var(phi, rhs, p.lam, p.Dxx, p.Dyy, p.Dzz, None, False)
            else:
                s.solve_const(phi, rhs, p.lam, p.Dc)
            return phi, s.history(), s.stats().staged_fields

    with capi.Solver(p.mx, p.delx, p.bc, L=2, eps=1e-8, nslabs=nslabs, slab_count=nslabs, devices=(0,)) as s:
        for t, (rhs, phi0) in enumerate(steps):
            want_phi, want_hist, want_staged = full(rhs, phi0)
            assert want_staged == (6 if p.var_coef else 3)
            phi = phi0.copy()
            if t == 0:
                if p.var_coef:
                    s.solve_var(phi, rhs, p.lam, p.Dxx, p.Dyy, p.Dzz, None, False)
                else:
                    s.solve_const(phi, rhs, p.lam, p.Dc)
            else:
                s.update_phi(phi0)
                s.update_rhs(rhs)
                assert s.resolve(phi) == 0
                assert s.stats().staged_fields == 2          # phi and rhs only: no coefficient crossed again
            assert np.array_equal(s.history(), want_hist) and len(want_hist) > 3
            assert np.array_equal(phi, want_phi)


def test_stepping_after_an_update_needs_a_reset():
    p = P.smooth_var(16)
    with capi.Solver(p.mx, p.delx, p.bc, L=1, eps=0.0) as s:
        with pytest.raises(capi.KryError):
            s.update_rhs(p.rhs)                              # nothing loaded yet
        s.load_var(p.phi, p.rhs, p.lam, p.Dxx, p.Dyy, p.Dzz, None, False)
        s.iterate(3)
        first = s.history().copy()
        s.update_rhs(2.0 * p.rhs)
        s.update_phi(p.phi)                                  # (phi on the device is the iterate of three passes)
        with pytest.raises(capi.KryError):
            s.iterate(1)
        s.reset()
        s.iterate(3)
        # BiCGStab is scale-invariant up to rounding: twice the right-hand side, twice the residuals
        assert np.allclose(s.history(), 2.0 * first, rtol=1e-9)
        s.update_lambda(p.lam + 5.0)
        s.update_phi(p.phi)
        s.reset()
        s.iterate(3)
        assert not np.allclose(s.history(), 2.0 * first, rtol=1e-3)


def test_adaptor_keeps_coefficients_between_solves():
    capi.lib()
    S = C.CDLL(os.path.join(ROOT, "krylov_amd", "libkrylov_compat_test.so"))
    i3 = C.c_int * 3
    S.compat_time_steps.restype = C.c_int
    S.compat_time_steps.argtypes = [i3, C.c_int, C.c_double, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp,
                                    C.POINTER(C.c_int), C.POINTER(C.c_int)]
    p = P.smooth_var(24)
    steps = 10
    res = {}
    for keep in (0, 1):
        phi = np.zeros((steps,) + p.phi.shape)
        iters, staged = (C.c_int * steps)(), (C.c_int * steps)()
        assert S.compat_time_steps(i3(*p.mx), 2, 1e-8, steps, keep, p.phi, phi, p.rhs, p.lam, p.Dxx, p.Dyy, p.Dzz,
                                   iters, staged) == 0
        res[keep] = (phi, list(iters), list(staged))
    assert res[0][2] == [6] * steps
    assert res[1][2] == [6] + [2] * (steps - 1)
    assert res[0][1] == res[1][1] and min(res[0][1]) > 3
    assert np.array_equal(res[0][0], res[1][0])


def test_adaptor_forgets_resident_coefficients_when_the_face_types_change():
    """keep-coefficients on, one face type changed between two solves on the same arrays: the adaptor
    builds a new device handle (which may land on the old one's address) and must stage all fields again."""
    capi.lib()
    S = C.CDLL(os.path.join(ROOT, "krylov_amd", "libkrylov_compat_test.so"))
    i3, i6 = C.c_int * 3, C.c_int * 6
    S.compat_keep_across_face_change.restype = C.c_int
    S.compat_keep_across_face_change.argtypes = [i3, C.c_int, C.c_double, i6, i6, _dp, _dp, _dp, _dp, _dp, _dp, _dp,
                                                 C.POINTER(C.c_int)]
    p = P.smooth_var(24)
    bc_a, bc_b = (0, 0, 0, 0, 0, 0), (0, 0, 0, 1, 0, 0)
    phi = np.zeros_like(p.phi)
    staged = (C.c_int * 2)()
    it = S.compat_keep_across_face_change(i3(*p.mx), 2, 1e-8, i6(*bc_a), i6(*bc_b), p.phi, phi, p.rhs, p.lam,
                                          p.Dxx, p.Dyy, p.Dzz, staged)
    assert list(staged) == [6, 6]
    with capi.Solver(p.mx, p.delx, bc_b, L=2, eps=1e-8) as s:
        want = p.phi.copy()
        s.solve_var(want, p.rhs, p.lam, p.Dxx, p.Dyy, p.Dzz, None, False)
        assert it == s.stats().iterations and it > 3
    assert np.array_equal(phi, want)


@pytest.mark.parametrize("bc", [(1, 0, 1, 1, 0, 1), (0,) * 6])
def test_fetch_after_a_finished_resident_solve_equals_the_full_solve(bc):
    """Constant coefficients: the reference's post-loop residual passes phi through do_BCs once more
    (solveLin_BICGStab.C:512-518, Krylov.C:516).  kry_resolve(NULL) + kry_fetch_phi, and
    kry_load + kry_iterate to the end + kry_fetch_phi, hand back the array kry_solve_const does --
    ghost and face cells included; a second fetch returns the same array (treated once)."""
    p = P.random_fields((20, 18, 22), 11, bc=bc, var_coef=False)
    rng = np.random.default_rng(2)
    phi0 = rng.uniform(-1, 1, p.phi.shape)
    with capi.Solver(p.mx, p.delx, p.bc, L=2, eps=1e-9, max_outer=400) as s:
        want = phi0.copy()
        s.solve_const(want, p.rhs, p.lam, p.Dc)
        n = s.stats().iterations
        assert n > 2
        s.update_phi(phi0)
        s.resolve(None)
        got = np.empty_like(want)
        s.fetch_phi(got)
        again = np.empty_like(want)
        s.fetch_phi(again)
        assert np.array_equal(got, want) and np.array_equal(again, want)
        s.load_const(phi0, p.rhs, p.lam, p.Dc)
        s.iterate(1)
        mid = np.empty_like(want)
        s.fetch_phi(mid)                     # unfinished: the untreated iterate
        s.iterate(10 * n)
        assert s.stats().iterations == n
        s.fetch_phi(got)
        assert np.array_equal(got, want) and not np.array_equal(mid, want)
