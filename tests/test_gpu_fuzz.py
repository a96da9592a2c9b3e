"""Randomised whole-solve parity: grid extents (odd, tiny, wider than a tile), spacings, face
types, coefficient kind, l, slab count and every field drawn from a seeded generator; the device
against the oracle summing in the device's order, bit for bit -- history, pass count, and the
returned array including boundary planes and ghost cells (one ghost corner exempt, see
test_start_vector_with_values_on_faces_and_ghosts in test_gpu_bitwise.py)."""
import numpy as np
import pytest

from helpers import oracle_solve, points
from krylov_amd import capi, problems as P

pytestmark = pytest.mark.gpu


def draw(seed):
    rng = np.random.default_rng(1000 + seed)
    wide = rng.integers(0, 6) == 0
    mx = (int(rng.integers(129, 150)) if wide else int(rng.integers(4, 41)), int(rng.integers(4, 41)), int(rng.integers(4, 33)))
    nslabs = 1
    if rng.integers(0, 3) == 0:
        for c in (4, 2):
            if mx[2] % c == 0 and mx[2] // c >= 4:
                nslabs = c
                break
    dyadic = rng.integers(0, 2) == 0
    delx = tuple((1.0 / 2 ** int(rng.integers(3, 7))) if dyadic else float(rng.uniform(0.01, 0.2)) for _ in range(3))
    bc = tuple(int(b) for b in rng.integers(0, 2, 6))
    if rng.integers(0, 4) == 0:
        bc = (0,) * 6
    kind = ("const", "var", "offdiag")[int(rng.integers(0, 3))]
    L = int(rng.integers(1, 5))
    shp = (mx[2] + 3, mx[1] + 3, mx[0] + 3)
    u = lambda lo, hi: rng.uniform(lo, hi, shp)
    p = P.Problem("fuzz %d" % seed, mx, delx, bc, kind != "const", kind == "offdiag",
                  tuple(float(v) for v in rng.uniform(0.5, 3.0, 3)), u(-1, 1), u(-1, 1), u(0, 2),
                  u(0.5, 2), u(0.5, 2), u(0.5, 2), u(-0.2, 0.2) if kind == "offdiag" else np.zeros(shp))
    return p, L, nslabs


@pytest.mark.parametrize("seed", range(200))
def test_random_configuration(orc, seed):
    check(orc, seed, False)


@pytest.mark.parametrize("seed", range(0, 200, 4))
def test_random_configuration_one_kernel_per_primitive(orc, seed):
    check(orc, seed, True)


@pytest.mark.parametrize("seed", range(200, 320))
def test_random_configuration_and_random_tiling(orc, monkeypatch, seed):
    """The same with the kernels' tiling drawn as well: chunk depth, tile rows, blocks of the streaming
    kernels, and the split launches of the overlapped exchange on or off."""
    rng = np.random.default_rng(5000 + seed)
    monkeypatch.setenv("KRY_ZC", str(int(rng.choice([2, 3, 5, 8, 13]))))
    monkeypatch.setenv("KRY_TY", str(int(rng.choice([8, 16]))))
    monkeypatch.setenv("KRY_STREAM_BLOCKS", str(int(rng.choice([64, 256, 2048]))))
    overlap = int(rng.integers(0, 2))
    monkeypatch.setenv("KRY_OVERLAP", str(overlap))
    check(orc, seed, False, split=bool(overlap))


def check(orc, seed, generic, split=True):
    p, L, nslabs = draw(seed)
    got = capi.solve_problem(p, L=L, eps=1e-13, max_outer=6, nslabs_local=nslabs, devices=(0,),
                             flags=capi.KRY_FORCE_GENERIC if generic else 0)
    ty, zc, sb = got["order"]
    orc.use_gpu_order({0: 2, 1: 1, 2: 3}[got["path"]], ty, zc, sb)
    orc.gpu_order_slabs(nslabs, split)
    try:
        want = oracle_solve(orc, p, L=L, eps=1e-13, max_outer=6)
    finally:
        orc.gpu_order_slabs(1, False)
        orc.use_gpu_order(0)
    what = "%s mx=%s bc=%s l=%d slabs=%d path=%d" % (p.name, p.mx, p.bc, L, nslabs, got["path"])
    assert (got["path"] == 0) if generic else (got["path"] in (1, 2)), what
    assert got["status"] == want["status"], what
    assert got["iters"] == want["iters"], what
    assert got["norm_rhs"] == want["norm_rhs"] and got["norm_init"] == want["norm_init"], what
    hw, hg = want["history"], got["history"]
    if np.all(np.isfinite(hw)):
        assert np.array_equal(hg, hw), what
        assert np.array_equal(points(got["phi"]), points(want["phi"])), what
        g, w = got["phi"].copy(), want["phi"].copy()
        if not generic:
            g[0, 0, 0] = w[0, 0, 0] = 0.
        assert np.array_equal(g, w), what
    else:
        assert np.array_equal(np.isfinite(hg), np.isfinite(hw)), what


@pytest.mark.parametrize("L", [1, 2, 3])
@pytest.mark.parametrize("kind", ["const", "var", "offdiag"])
@pytest.mark.parametrize("bc", [(0,) * 6, (1,) * 6, (0, 1, 1, 0, 1, 0)])
@pytest.mark.parametrize("mx", [(4, 4, 4), (4, 5, 6), (6, 4, 5), (5, 9, 4), (130, 4, 4)])
def test_smallest_grids(orc, mx, bc, kind, L):
    """Four intervals per direction is the smallest grid the library takes."""
    rng = np.random.default_rng(7)
    shp = (mx[2] + 3, mx[1] + 3, mx[0] + 3)
    u = lambda lo, hi: rng.uniform(lo, hi, shp)
    p = P.Problem("tiny", mx, (0.1, 0.2, 0.25), bc, kind != "const", kind == "offdiag", (1., 2., 3.), u(-1, 1), u(-1, 1), u(0, 2),
                  u(.5, 2), u(.5, 2), u(.5, 2), u(-.2, .2) if kind == "offdiag" else np.zeros(shp))
    got = capi.solve_problem(p, L=L, eps=1e-13, max_outer=4)
    ty, zc, sb = got["order"]
    orc.use_gpu_order({0: 2, 1: 1, 2: 3}[got["path"]], ty, zc, sb)
    try:
        want = oracle_solve(orc, p, L=L, eps=1e-13, max_outer=4)
    finally:
        orc.use_gpu_order(0)
    assert got["iters"] == want["iters"] and got["status"] == want["status"]
    hw, hg = want["history"], got["history"]
    assert np.array_equal(np.isfinite(hw), np.isfinite(hg))
    if np.all(np.isfinite(hw)):
        assert np.array_equal(hw, hg) and np.array_equal(points(got["phi"]), points(want["phi"]))


def test_grids_below_four_intervals_are_refused():
    with pytest.raises(capi.KryError, match="bad grid"):
        capi.Solver((3, 8, 8), (0.1, 0.1, 0.1), (0,) * 6, L=2)
