"""Bit-for-bit parity of complete solves.

The device path evaluates every stencil value and every vector update with the
reference's roundings (checked per operator in test_gpu_ops.py) and differs from
the reference's serial code only in the ORDER in which dot products are summed.
oracle/gpu_order.c repeats exactly that order on the CPU.  With it installed the
oracle and the device must agree on every number of the solve: the residual
history of all passes (50 and more), the pass count, and the solution.

This is the strongest form of the required parity (history <= 1e-10 relative
over the first 50 passes, solution <= 1e-9 relative L2): here both are 0.  The
same oracle in the reference's own summation order drifts away from either of
them chaotically; test_summation_order_alone_moves_the_history measures that."""
import numpy as np
import pytest

from helpers import oracle_solve, points, rel_l2
from krylov_amd import capi, problems as P

pytestmark = pytest.mark.gpu


def make(kind, n):
    rng = np.random.default_rng(17)
    if kind == "ref_test1":
        return P.reference_test1(n)
    if kind == "const_random_rhs":
        p = P.reference_test1(n)
        p.rhs = P.interior_only(rng.uniform(-1, 1, p.rhs.shape))
        return p
    if kind == "ref_test2":
        return P.reference_test2(n)
    if kind == "ref_test3":
        return P.reference_test3(n)
    if kind == "smooth_var":
        return P.smooth_var(n)
    if kind == "var_random_rhs":
        p = P.smooth_var(n)
        p.rhs = rng.uniform(-1, 1, p.rhs.shape)
        return p
    if kind == "offdiag":
        return P.offdiag_var(n)
    raise KeyError(kind)


def run_pair(orc, p, L, eps, max_outer, flags=0, nslabs=1, split=False):
    got = capi.solve_problem(p, L=L, eps=eps, max_outer=max_outer, flags=flags, nslabs_local=nslabs, devices=(0,))
    ty, zc, sb = got["order"]
    orc.use_gpu_order({0: 2, 1: 1, 2: 3}[got["path"]], ty, zc, sb)
    orc.gpu_order_slabs(nslabs, split)
    try:
        want = oracle_solve(orc, p, L=L, eps=eps, max_outer=max_outer)
    finally:
        orc.gpu_order_slabs(1, False)
        orc.use_gpu_order(0)
    return want, got


def assert_identical(want, got, min_passes):
    assert got["status"] == want["status"]
    assert got["iters"] == want["iters"] >= min_passes
    assert got["norm_rhs"] == want["norm_rhs"]
    assert got["norm_init"] == want["norm_init"]
    hw, hg = want["history"], got["history"]
    assert len(hg) == len(hw)
    # the norms themselves pass through one sqrt on either side
    assert np.max(np.abs(hg - hw) / want["norm_rhs"]) <= 1e-15
    assert np.array_equal(points(got["phi"]), points(want["phi"]))


@pytest.mark.parametrize("L", [1, 2])
@pytest.mark.parametrize("kind,n", [("ref_test1", 32), ("const_random_rhs", 32), ("const_random_rhs", 64),
                                    ("ref_test2", 32), ("ref_test3", 32), ("smooth_var", 32),
                                    ("var_random_rhs", 48), ("offdiag", 32)])
def test_fused_path_matches_oracle_in_device_order(orc, kind, n, L):
    p = make(kind, n)
    want, got = run_pair(orc, p, L, eps=1e-13, max_outer=60)
    assert got["fused"]
    assert_identical(want, got, min_passes=50 if "random" in kind else 10)


@pytest.mark.parametrize("L", [1, 2])
def test_fused_path_to_convergence(orc, L):
    p = make("var_random_rhs", 32)
    want, got = run_pair(orc, p, L, eps=1e-6, max_outer=0)
    assert_identical(want, got, min_passes=20)


@pytest.mark.parametrize("L", [1, 2, 4])
@pytest.mark.parametrize("bc", [(0,) * 6, (0, 0, 1, 1, 0, 1), (1, 0, 0, 1, 1, 0)])
def test_generic_path_matches_oracle_in_device_order(orc, bc, L):
    p = P.smooth_var(24, bc=bc)
    p.rhs = np.random.default_rng(3).uniform(-1, 1, p.rhs.shape)
    want, got = run_pair(orc, p, L, eps=1e-13, max_outer=50, flags=capi.KRY_FORCE_GENERIC)
    assert not got["fused"]
    assert_identical(want, got, min_passes=50 // L if L > 2 else 50)


@pytest.mark.parametrize("overlap", ["0", "1"])
@pytest.mark.parametrize("nslabs", [2, 4])
@pytest.mark.parametrize("kind,n,L", [("const_random_rhs", 32, 1), ("const_random_rhs", 32, 2), ("var_random_rhs", 48, 2),
                                      ("offdiag", 32, 1), ("var_random_rhs", (40, 24, 64), 2)])
def test_slab_runs_match_oracle_in_device_order(orc, monkeypatch, kind, n, L, nslabs, overlap):
    """Several z slabs (per-slab sums added in slab order; with KRY_OVERLAP=1 and short
    chunks also the edge/interior split of the stencil launches): still bit for bit."""
    monkeypatch.setenv("KRY_OVERLAP", overlap)
    monkeypatch.setenv("KRY_ZC", "4")
    if isinstance(n, tuple):
        rng = np.random.default_rng(17)
        p = P.smooth_var(n)
        p.rhs = rng.uniform(-1, 1, p.rhs.shape)
    else:
        p = make(kind, n)
    want, got = run_pair(orc, p, L, eps=1e-13, max_outer=40, nslabs=nslabs, split=(overlap == "1"))
    assert got["fused"]
    assert_identical(want, got, min_passes=10)


@pytest.mark.parametrize("nslabs", [2, 4])
@pytest.mark.parametrize("L", [1, 3])
def test_generic_path_slab_runs_match_oracle_in_device_order(orc, nslabs, L):
    p = P.smooth_var(24, bc=(0, 0, 1, 1, 0, 1))
    p.rhs = np.random.default_rng(3).uniform(-1, 1, p.rhs.shape)
    want, got = run_pair(orc, p, L, eps=1e-13, max_outer=30, flags=capi.KRY_FORCE_GENERIC, nslabs=nslabs)
    assert got["path"] == 0
    assert_identical(want, got, min_passes=10)


@pytest.mark.parametrize("nslabs", [2, 4])
@pytest.mark.parametrize("L", [1, 2])
@pytest.mark.parametrize("bc", [(0, 0, 1, 1, 0, 1), (1, 1, 0, 1, 1, 1)])
def test_general_boundary_slab_runs_match_oracle_in_device_order(orc, bc, L, nslabs):
    p = P.smooth_var((24, 20, 32), bc=bc)
    p.rhs = np.random.default_rng(3).uniform(-1, 1, p.rhs.shape)
    want, got = run_pair(orc, p, L, eps=1e-13, max_outer=30, nslabs=nslabs)
    assert got["path"] == 2
    assert_identical(want, got, min_passes=10)


MIXED = [(0, 0, 1, 1, 0, 1), (1, 0, 0, 1, 1, 0), (1, 1, 1, 1, 1, 1), (0, 0, 0, 0, 1, 1), (0, 1, 0, 0, 0, 0)]


@pytest.mark.parametrize("L", [1, 2])
@pytest.mark.parametrize("bc", MIXED)
def test_general_boundary_fused_path_matches_oracle_in_device_order(orc, bc, L):
    """Extrapolating / mixed faces with field coefficients run the fused kernels'
    general-boundary variant (vector updates on ghost cells, boundary treatment
    and boundary-plane dot terms after each stencil kernel)."""
    p = P.smooth_var(24, bc=bc)
    p.rhs = np.random.default_rng(3).uniform(-1, 1, p.rhs.shape)
    want, got = run_pair(orc, p, L, eps=1e-13, max_outer=50)
    assert got["path"] == 2
    assert_identical(want, got, min_passes=50)


@pytest.mark.parametrize("L", [1, 2])
@pytest.mark.parametrize("bc", MIXED + [(1, 1, 1, 0, 0, 1), (0, 1, 1, 1, 1, 0)])
def test_constant_coefficients_on_general_faces_fused_path(orc, bc, L):
    """The constant-coefficient operator passes its INPUT through do_BCs (Krylov.C:516), which
    rewrites boundary planes and ghost cells of the iteration vectors before every application.
    The fused general-boundary kernels evaluate that treatment cell by cell while they form the
    input; against the oracle, which applies do_BCs as the reference does."""
    p = P.reference_test1(24)
    p.bc = bc
    p.rhs = np.random.default_rng(13).uniform(-1, 1, p.rhs.shape)
    p.phi = 1e-2 * np.random.default_rng(14).uniform(-1, 1, p.phi.shape)     # a start vector that do_BCs changes
    want, got = run_pair(orc, p, L, eps=1e-13, max_outer=40)
    assert got["path"] == 2
    assert_identical(want, got, min_passes=40)


@pytest.mark.parametrize("L", [1, 2])
@pytest.mark.parametrize("n", [(40, 20, 28), (130, 12, 9), (16, 62, 16), (128, 8, 8), (4, 4, 4), (5, 4, 6)])
def test_constant_coefficients_on_general_faces_ragged_shapes(orc, n, L):
    """mx a multiple of the tile width (the upper x face then lies in the halo column), several
    tiles per row, tile edges inside the ghost layer, chunks of a few planes; the smallest boxes
    the library takes (every cell but a few lies on or behind a face)."""
    q = P.smooth_var(n, bc=(1, 1, 1, 1, 1, 1))
    p = P.Problem("const", q.mx, q.delx, q.bc, False, False, (1., 2., 3.), q.phi, q.rhs, q.lam)
    p.rhs = np.random.default_rng(15).uniform(-1, 1, p.rhs.shape)
    want, got = run_pair(orc, p, L, eps=1e-13, max_outer=25)
    assert got["path"] == 2
    assert_identical(want, got, min_passes=25)


@pytest.mark.parametrize("nslabs", [2, 4])
def test_constant_coefficients_on_general_faces_slabs(orc, nslabs):
    p = P.reference_test1((24, 20, 32))
    p.bc = (1, 0, 1, 1, 1, 1)
    p.rhs = np.random.default_rng(16).uniform(-1, 1, p.rhs.shape)
    want, got = run_pair(orc, p, 2, eps=1e-13, max_outer=25, nslabs=nslabs)
    assert got["path"] == 2
    assert_identical(want, got, min_passes=10)


@pytest.mark.parametrize("L", [1, 2])
@pytest.mark.parametrize("n", [(40, 20, 28), (130, 12, 9), (16, 62, 16)])
def test_general_boundary_fused_path_ragged_shapes(orc, n, L):
    """Tile edges that fall inside the ghost layer, several tiles per row, short chunks."""
    bc = (1, 1, 0, 1, 1, 1)
    p = P.smooth_var(n, bc=bc)
    p.rhs = np.random.default_rng(5).uniform(-1, 1, p.rhs.shape)
    want, got = run_pair(orc, p, L, eps=1e-13, max_outer=30)
    assert got["path"] == 2
    assert_identical(want, got, min_passes=30)


@pytest.mark.parametrize("L", [1, 2, 3, 4])
def test_general_boundary_fused_path_with_offdiagonal_term(orc, L):
    """The 11-point operator reads edge ghosts, which only the whole-array update
    semantics and the face order of do_BCs define."""
    p = P.offdiag_var(24)
    p.bc = (1, 0, 1, 1, 0, 1)
    p.rhs = np.random.default_rng(8).uniform(-1, 1, p.rhs.shape)
    want, got = run_pair(orc, p, L, eps=1e-13, max_outer=30)
    assert got["path"] == 2
    assert_identical(want, got, min_passes=30)


@pytest.mark.parametrize("L", [1, 2, 3])
@pytest.mark.parametrize("bc", [(1, 1, 1, 1, 1, 1), (1, 0, 1, 1, 0, 1), (1, 1, 1, 0, 1, 1)])
def test_edge_ghosts_with_offdiagonal_term_away_from_zero(orc, bc, L):
    """Dxy that does not vanish on the faces: the operator value at an edge point then depends on
    the xy edge ghost of its input, which the x-face treatment of an earlier operator output
    derived from whatever the ghost rows of that array held at the time (do_BCs treats x first,
    over the full tangential range, BoundaryHandler.C:420-421).  The device must keep every
    array's ghost cells in step with the reference's arrays for this to come out equal."""
    p = P.offdiag_var((24, 20, 28))
    p.bc = bc
    rng = np.random.default_rng(21)
    p.Dxy = 0.05 + 0.1 * rng.uniform(0, 1, p.Dxy.shape)
    p.rhs = rng.uniform(-1, 1, p.rhs.shape)
    want, got = run_pair(orc, p, L, eps=1e-13, max_outer=20, flags=capi.KRY_FORCE_GENERIC)
    assert_identical(want, got, min_passes=5)
    want, got = run_pair(orc, p, L, eps=1e-13, max_outer=20)
    assert got["path"] == 2
    assert_identical(want, got, min_passes=5)
    want, got = run_pair(orc, p, L, eps=1e-13, max_outer=20, nslabs=2, split=True)
    assert got["path"] == 2
    assert_identical(want, got, min_passes=5)


@pytest.mark.parametrize("L", [1, 2])
def test_high_contrast_mixed_faces_workload(orc, L):
    """The benchmark's anisotropic high-contrast problem (coefficient contrast 1e4,
    x faces Dirichlet, y faces and the upper z face extrapolating) at 48^3."""
    import bench
    f = bench.slab_problem(48, 1, 0, "mixed")
    p = P.Problem("mixed", f["mx"], f["delx"], f["bc"], True, False, (1., 1., 1.), f["phi"], f["rhs"], f["lam"],
                  f["Dxx"], f["Dyy"], f["Dzz"], np.zeros_like(f["phi"]))
    want, got = run_pair(orc, p, L, eps=1e-13, max_outer=40)
    assert got["path"] == 2
    assert_identical(want, got, min_passes=40)


@pytest.mark.parametrize("L", [1, 2, 3])
@pytest.mark.parametrize("kind,bc", [("var", (0,) * 6), ("offdiag", (0,) * 6), ("const", (0,) * 6),
                                     ("var", (1, 0, 1, 1, 0, 1)), ("offdiag", (1, 1, 0, 1, 1, 1)), ("const", (0, 1, 1, 0, 1, 1))])
def test_start_vector_with_values_on_faces_and_ghosts(orc, kind, bc, L):
    """A caller's phi that is not zero on the boundary planes and ghost cells: the field-coefficient
    operator reads it as it stands (no input treatment, Krylov.C:407-496), the constant-coefficient
    one treats it first (Krylov.C:516), and the whole-array updates carry its boundary values along."""
    rng = np.random.default_rng(31)
    if kind == "const":
        p = P.reference_test1((24, 20, 28))
        p.bc = bc
    else:
        p = (P.offdiag_var if kind == "offdiag" else P.smooth_var)((24, 20, 28), bc=bc)
        if kind == "offdiag":
            p.Dxy = 0.05 + 0.1 * rng.uniform(0, 1, p.Dxy.shape)
    p.rhs = rng.uniform(-1, 1, p.rhs.shape)
    p.phi = rng.uniform(-1, 1, p.phi.shape)
    want, got = run_pair(orc, p, L, eps=1e-13, max_outer=12)
    assert got["fused"]
    assert_identical(want, got, min_passes=5)
    # ghost cells and boundary planes of the returned array too.  One cell is exempt: the ghost corner
    # (-1,-1,-1), which no operator reads; the device's streaming kernels start behind it and return the
    # caller's value there, the reference carries the whole-array updates through it.
    g, w = got["phi"].copy(), want["phi"].copy()
    g[0, 0, 0] = w[0, 0, 0] = 0.
    assert np.array_equal(g, w)
    # two slabs, and one kernel per primitive (which has no exempt cell)
    want, got = run_pair(orc, p, L, eps=1e-13, max_outer=12, nslabs=2, split=True)   # (overlapped exchange is the default)
    assert_identical(want, got, min_passes=5)
    want, got = run_pair(orc, p, L, eps=1e-13, max_outer=12, flags=capi.KRY_FORCE_GENERIC)
    assert_identical(want, got, min_passes=5)
    assert np.array_equal(got["phi"], want["phi"])


# ---- l > 2: one stencil kernel pair per BiCG step, one streaming kernel per Gram-Schmidt dot ----

@pytest.mark.parametrize("L", [3, 4, 6])
@pytest.mark.parametrize("kind,n", [("const_random_rhs", 32), ("var_random_rhs", 48), ("offdiag", 32), ("ref_test2", 32)])
def test_fused_path_any_l(orc, kind, n, L):
    p = make(kind, n)
    want, got = run_pair(orc, p, L, eps=1e-13, max_outer=24)
    assert got["path"] == 1
    assert_identical(want, got, min_passes=3)


def test_fused_path_largest_l(orc):
    p = make("var_random_rhs", 32)
    want, got = run_pair(orc, p, 8, eps=1e-13, max_outer=6)
    assert got["path"] == 1
    assert_identical(want, got, min_passes=3)


@pytest.mark.parametrize("L", [3, 4])
def test_fused_path_any_l_to_convergence(orc, L):
    p = make("var_random_rhs", 32)
    want, got = run_pair(orc, p, L, eps=1e-6, max_outer=0)
    assert got["status"] == 0
    assert_identical(want, got, min_passes=5)


@pytest.mark.parametrize("L", [3, 4])
@pytest.mark.parametrize("bc", [(0, 0, 1, 1, 0, 1), (1, 1, 1, 1, 1, 1), (0, 1, 0, 0, 0, 0)])
def test_general_boundary_fused_path_any_l(orc, bc, L):
    p = P.smooth_var(24, bc=bc)
    p.rhs = np.random.default_rng(3).uniform(-1, 1, p.rhs.shape)
    want, got = run_pair(orc, p, L, eps=1e-13, max_outer=20)
    assert got["path"] == 2
    assert_identical(want, got, min_passes=3)


@pytest.mark.parametrize("L", [3, 5])
@pytest.mark.parametrize("bc", [(1, 0, 0, 1, 1, 0), (1, 1, 1, 0, 0, 1)])
def test_constant_coefficients_on_general_faces_any_l(orc, bc, L):
    p = P.reference_test1((40, 20, 28))
    p.bc = bc
    p.rhs = np.random.default_rng(13).uniform(-1, 1, p.rhs.shape)
    p.phi = 1e-2 * np.random.default_rng(14).uniform(-1, 1, p.phi.shape)
    want, got = run_pair(orc, p, L, eps=1e-13, max_outer=15)
    assert got["path"] == 2
    assert_identical(want, got, min_passes=3)


@pytest.mark.parametrize("overlap", ["0", "1"])
@pytest.mark.parametrize("nslabs", [2, 4])
@pytest.mark.parametrize("kind,n,L", [("const_random_rhs", 32, 3), ("var_random_rhs", (40, 24, 64), 4)])
def test_slab_runs_any_l(orc, monkeypatch, kind, n, L, nslabs, overlap):
    monkeypatch.setenv("KRY_OVERLAP", overlap)
    monkeypatch.setenv("KRY_ZC", "4")
    if isinstance(n, tuple):
        p = P.smooth_var(n)
        p.rhs = np.random.default_rng(17).uniform(-1, 1, p.rhs.shape)
    else:
        p = make(kind, n)
    want, got = run_pair(orc, p, L, eps=1e-13, max_outer=15, nslabs=nslabs, split=(overlap == "1"))
    assert got["path"] == 1
    assert_identical(want, got, min_passes=3)


@pytest.mark.parametrize("nslabs", [2, 4])
def test_general_boundary_slab_runs_any_l(orc, nslabs):
    p = P.smooth_var((24, 20, 32), bc=(1, 1, 0, 1, 1, 1))
    p.rhs = np.random.default_rng(3).uniform(-1, 1, p.rhs.shape)
    want, got = run_pair(orc, p, 3, eps=1e-13, max_outer=15, nslabs=nslabs)
    assert got["path"] == 2
    assert_identical(want, got, min_passes=3)


def test_summation_order_alone_moves_the_history(orc):
    """Reference order against device order, both on the CPU oracle: identical
    arithmetic, different summation order.  The histories agree at the start and
    separate by far more than 1e-10 within 50 passes -- a property of
    BiCGStab(l), not of either implementation."""
    p = make("var_random_rhs", 32)
    ref = oracle_solve(orc, p, L=2, eps=1e-13, max_outer=50)
    orc.use_gpu_order(1)
    try:
        dev = oracle_solve(orc, p, L=2, eps=1e-13, max_outer=50)
    finally:
        orc.use_gpu_order(0)
    rel = np.abs(ref["history"] - dev["history"]) / ref["norm_rhs"]
    assert rel[0] <= 1e-13
    assert rel[:3].max() <= 1e-12
    assert rel.max() > 1e-10
