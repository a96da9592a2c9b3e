"""Full-size checks at 256^3, 512^3 and on the 1024 x 1024 x 128 block (the sizes of
the benchmark configs).  A converged solve of the CPU oracle would take minutes to
hours there: the first passes are compared with the oracle bit for bit, the rest
are size-independent properties and cross-checks between the library's two device
implementations."""
import numpy as np
import pytest

from helpers import oracle_solve, points, rel_l2
from krylov_amd import capi, problems as P

pytestmark = pytest.mark.gpu

N = 256


@pytest.fixture(scope="module")
def prob():
    p = P.smooth_var(N)
    p.rhs = np.random.default_rng(1).uniform(-1, 1, p.rhs.shape)
    return p


def test_fused_and_generic_operator_agree_bit_for_bit(prob):
    p = prob
    psi = np.random.default_rng(2).uniform(-1, 1, p.rhs.shape)
    with capi.Solver(p.mx, p.delx, p.bc) as s:
        a = s.op_spmv_var(psi, p.lam, p.Dxx, p.Dyy, p.Dzz, None, False, fused=True)
        b = s.op_spmv_var(psi, p.lam, p.Dxx, p.Dyy, p.Dzz, None, False, fused=False)
    assert np.array_equal(points(a), points(b))
    assert np.all(np.isfinite(a)) and np.abs(points(a)).max() > 0


def test_operator_linearity(prob):
    p = prob
    rng = np.random.default_rng(3)
    x, y = rng.uniform(-1, 1, p.rhs.shape), rng.uniform(-1, 1, p.rhs.shape)
    with capi.Solver(p.mx, p.delx, p.bc) as s:
        f = lambda v: s.op_spmv_var(v, p.lam, p.Dxx, p.Dyy, p.Dzz, None, False, fused=True)
        lhs, rhs = f(x + 2.0 * y), f(x) + 2.0 * f(y)
    assert np.max(np.abs(lhs - rhs)) <= 1e-10 * np.max(np.abs(rhs))


def test_slabs_agree_with_single_slab_at_full_size(prob):
    one = capi.solve_problem(prob, L=2, eps=1e-30, max_outer=3)
    four = capi.solve_problem(prob, L=2, eps=1e-30, max_outer=3, nslabs_local=4, devices=(0,))
    assert one["fused"] and four["fused"]
    assert np.max(np.abs(one["history"] - four["history"])) <= 1e-11 * one["norm_rhs"]
    assert rel_l2(points(four["phi"]), points(one["phi"])) <= 1e-11


@pytest.mark.parametrize("L", [1, 2])
def test_converged_solution_has_small_true_residual(prob, L):
    p = prob
    got = capi.solve_problem(p, L=L, eps=1e-6)
    assert got["status"] == 0 and got["fused"]
    assert got["history"][-1] <= 1e-6 * got["norm_rhs"]
    # residual recomputed with the generic (non-fused) operator kernel
    with capi.Solver(p.mx, p.delx, p.bc) as s:
        ax = s.op_spmv_var(got["phi"], p.lam, p.Dxx, p.Dyy, p.Dzz, None, False, apply_bcs=False, fused=False)
        r = P.interior_only(p.rhs - ax)
        r[1, :, :] = r[-2, :, :] = 0
        r[:, 1, :] = r[:, -2, :] = 0
        r[:, :, 1] = r[:, :, -2] = 0
        nrm = s.op_l2norm(r)
    assert nrm <= 2e-6 * got["norm_rhs"]


def test_weak_scaling_block_shape_1024x1024x128():
    """The per-GPU block of the 1024^3 weak-scaling configuration: widest rows the
    configs use (8 tiles per row).  Fused against generic operator, bit for bit,
    and a few passes of the solve."""
    import bench
    f = bench.slab_problem(1024, 1, 0, "var", nz=128)
    psi = np.random.default_rng(4).uniform(-1, 1, f["rhs"].shape)
    with capi.Solver(f["mx"], f["delx"], (0,) * 6, L=2, eps=0.0) as s:
        a = s.op_spmv_var(psi, f["lam"], f["Dxx"], f["Dyy"], f["Dzz"], None, False, fused=True)
        b = s.op_spmv_var(psi, f["lam"], f["Dxx"], f["Dyy"], f["Dzz"], None, False, fused=False)
        assert np.array_equal(points(a), points(b))
        del a, b
        s.load_var(f["phi"], f["rhs"], f["lam"], f["Dxx"], f["Dyy"], f["Dzz"], None, False)
        assert s.iterate(5) == 0
        h = s.history()
    assert len(h) == 5 and np.all(np.isfinite(h)) and h[-1] < h[0]


@pytest.mark.parametrize("workload,passes,tol", [("var", 3, 1e-9), ("mixed", 1, 1e-9), ("mixed", 2, 1e-5)])
def test_benchmark_grid_512_fused_against_one_kernel_per_primitive(workload, passes, tol):
    """The benchmark's 512^3 problems: the fused kernels (Dirichlet faces / mixed
    faces) against the library's other device implementation, which evaluates the
    reference's statements one kernel at a time.  The two differ in the summation
    order of the dot products only, so the first passes agree to rounding noise
    (later passes separate chaotically; on the high-contrast problem, whose dot products cancel to
    eight digits, the second pass already differs by 1e-7 for some chunk depths)."""
    import bench
    f = bench.slab_problem(512, 1, 0, workload)
    hist = []
    for flags, path in ((0, 1 if workload == "var" else 2), (capi.KRY_FORCE_GENERIC, 0)):
        with capi.Solver(f["mx"], f["delx"], f["bc"], L=2, eps=0.0, flags=flags) as s:
            s.load_var(f["phi"], f["rhs"], f["lam"], f["Dxx"], f["Dyy"], f["Dzz"], None, False)
            assert s.iterate(passes) == 0
            assert s.stats().fused == path
            hist.append(s.history())
    a, b = hist
    assert len(a) == len(b) == passes
    assert np.max(np.abs(a - b) / np.abs(b)) <= tol


@pytest.mark.parametrize("overlap", ["1", "0"])
def test_eight_gpu_slab_shape_split_launch(monkeypatch, overlap):
    """The slab an 8-GPU run of the 512^3 benchmark gives each GPU (512 x 512 x 64):
    chunking, edge/interior split of the stencil kernels and the overlapped exchange
    as they will run there, here with two such slabs against the undivided 512 x 512 x 128 grid."""
    import bench
    monkeypatch.setenv("KRY_OVERLAP", overlap)
    f = bench.slab_problem(512, 1, 0, "var", nz=128)
    hist = []
    for ns in (1, 2):
        with capi.Solver(f["mx"], f["delx"], f["bc"], L=2, eps=0.0, nslabs=ns, slab_count=ns, devices=(0,)) as s:
            s.load_var(f["phi"], f["rhs"], f["lam"], f["Dxx"], f["Dyy"], f["Dzz"], None, False)
            assert s.iterate(4) == 0 and s.stats().fused == 1
            hist.append(s.history())
            assert s.stats().chunk_planes == (17 if ns == 2 else s.stats().chunk_planes)   # 65 planes in 4 chunks
    # (summation order differs between the two layouts; by the fourth pass that is a few 1e-10)
    assert np.max(np.abs(hist[0] - hist[1]) / hist[0]) <= 1e-9
    assert abs(hist[0][0] - hist[1][0]) / hist[0][0] <= 1e-12


def _bench_problem(n, workload):
    import bench
    f = bench.slab_problem(n, 1, 0, "const" if workload == "const_faces" else workload)
    if workload == "const_faces":
        # constant coefficients on mixed faces: the operator's input passes through do_BCs (Krylov.C:516)
        return P.Problem("const", f["mx"], f["delx"], (0, 0, 1, 1, 0, 1), False, False, tuple(f["Dc"]), f["phi"], f["rhs"],
                         f["lam"])
    if workload == "const":
        return P.Problem("const", f["mx"], f["delx"], f["bc"], False, False, tuple(f["Dc"]), f["phi"], f["rhs"], f["lam"])
    return P.Problem(workload, f["mx"], f["delx"], f["bc"], True, False, (1., 1., 1.), f["phi"], f["rhs"], f["lam"],
                     f["Dxx"], f["Dyy"], f["Dzz"], np.zeros_like(f["phi"]))


@pytest.mark.parametrize("n,workload,L,passes", [(256, "const", 1, 4), (256, "const", 2, 2), (256, "mixed", 2, 2), (256, "const_faces", 2, 2),
                                                 (512, "var", 2, 1), (512, "var", 1, 1), (512, "mixed", 2, 1)])
def test_benchmark_grids_first_passes_against_the_oracle_bit_for_bit(orc, n, workload, L, passes):
    """The benchmark's own problems at their own sizes (256^3 constant coefficients, the 256^3
    high-contrast mixed-face problem, 512^3 field coefficients): the first passes of the device
    solve and of the oracle summing in the device's order carry the same bits."""
    p = _bench_problem(n, workload)
    got = capi.solve_problem(p, L=L, eps=0.0, max_outer=passes)
    ty, zc, sb = got["order"]
    orc.use_gpu_order({1: 1, 2: 3}[got["path"]], ty, zc, sb)
    try:
        want = oracle_solve(orc, p, L=L, eps=0.0, max_outer=passes)
    finally:
        orc.use_gpu_order(0)
    assert got["iters"] == want["iters"] == passes
    assert got["norm_rhs"] == want["norm_rhs"] and got["norm_init"] == want["norm_init"]
    assert np.max(np.abs(got["history"] - want["history"]) / want["norm_rhs"]) <= 1e-15
    assert np.array_equal(points(got["phi"]), points(want["phi"]))


def test_weak_scaling_block_first_pass_against_the_oracle_bit_for_bit(orc):
    """The 1024 x 1024 x 128 block a GPU holds in the weak-scaling configuration (eight tiles per
    row, rows of 1040 doubles): one pass of the l = 2 solve, device against oracle in device order."""
    import bench
    f = bench.slab_problem(1024, 1, 0, "var", nz=128)
    p = P.Problem("weak block", f["mx"], f["delx"], f["bc"], True, False, (1., 1., 1.), f["phi"], f["rhs"], f["lam"],
                  f["Dxx"], f["Dyy"], f["Dzz"], np.zeros_like(f["phi"]))
    got = capi.solve_problem(p, L=2, eps=0.0, max_outer=1)
    ty, zc, sb = got["order"]
    orc.use_gpu_order(1, ty, zc, sb)
    try:
        want = oracle_solve(orc, p, L=2, eps=0.0, max_outer=1)
    finally:
        orc.use_gpu_order(0)
    assert got["iters"] == want["iters"] == 1
    assert got["norm_rhs"] == want["norm_rhs"] and got["norm_init"] == want["norm_init"]
    assert np.array_equal(got["history"], want["history"])
    assert np.array_equal(points(got["phi"]), points(want["phi"]))
