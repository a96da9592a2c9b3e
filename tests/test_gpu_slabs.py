"""z-slab decomposition on one device: several slabs of one process share the
GPU and exchange their interface planes with device-to-device copies.  The
arithmetic per grid point is that of the single-slab solve; only the dot
products are summed slab by slab."""
import numpy as np
import pytest

from helpers import oracle_solve, points, rel_l2
from krylov_amd import capi, problems as P

pytestmark = pytest.mark.gpu


def cfg_of(orc, p):
    return orc.make_cfg(p.mx, p.delx, p.bc, var_coef=p.var_coef, use_offdiag=p.use_offdiag, Dc=p.Dc)


@pytest.mark.parametrize("nslabs", [2, 4])
@pytest.mark.parametrize("fused", [False, True])
def test_spmv_across_slabs_bit_exact(orc, nslabs, fused):
    p = P.random_fields((24, 20, 32), seed=21, var_coef=True, use_offdiag=True)
    want = orc.spmv_var(cfg_of(orc, p), p.phi, p.lam, p.Dxx, p.Dyy, p.Dzz, p.Dxy, True)
    with capi.Solver(p.mx, p.delx, p.bc, nslabs=nslabs, slab_count=nslabs, devices=(0,)) as s:
        got = s.op_spmv_var(p.phi, p.lam, p.Dxx, p.Dyy, p.Dzz, p.Dxy, True, apply_bcs=True, fused=fused)
    assert np.array_equal(points(got), points(want))


@pytest.mark.parametrize("nslabs", [2, 4])
def test_dot_across_slabs(orc, nslabs):
    p = P.random_fields((24, 20, 32), seed=22)
    cfg = cfg_of(orc, p)
    with capi.Solver(p.mx, p.delx, p.bc, nslabs=nslabs, slab_count=nslabs, devices=(0,)) as s:
        d, n = s.op_dot(p.phi, p.rhs), s.op_l2norm(p.phi)
    assert d == pytest.approx(orc.dot(cfg, p.phi, p.rhs), rel=1e-12)
    assert n == pytest.approx(orc.l2norm(cfg, p.phi), rel=1e-13)


@pytest.mark.parametrize("L", [1, 2])
@pytest.mark.parametrize("nslabs", [2, 4])
@pytest.mark.parametrize("prob", ["const", "var", "offdiag"])
def test_first_passes_match_single_slab(orc, prob, nslabs, L):
    """Three passes from the same start: any error in the ghost-plane handling
    of the fused updates would show at order one; rounding noise stays tiny."""
    if prob == "const":
        p = P.reference_test1(32)
        rng = np.random.default_rng(5)
        p.rhs = P.interior_only(rng.uniform(-1, 1, p.rhs.shape))
    elif prob == "var":
        p = P.smooth_var(32)
        p.rhs = np.random.default_rng(6).uniform(-1, 1, p.rhs.shape)
    else:
        p = P.offdiag_var(32)
        p.rhs = np.random.default_rng(7).uniform(-1, 1, p.rhs.shape)
    one = capi.solve_problem(p, L=L, eps=1e-30, max_outer=3)
    many = capi.solve_problem(p, L=L, eps=1e-30, max_outer=3, nslabs_local=nslabs, devices=(0,))
    want = oracle_solve(orc, p, L=L, eps=1e-30, max_outer=3)
    assert one["fused"] and many["fused"]
    assert many["iters"] == one["iters"] == 3
    for a in (one, many):
        assert np.max(np.abs(a["history"] - want["history"])) <= 1e-11 * want["norm_rhs"]
        assert rel_l2(points(a["phi"]), points(want["phi"])) <= 1e-11


@pytest.mark.parametrize("nslabs", [2, 4])
def test_full_solve_on_slabs(orc, nslabs):
    p = P.smooth_var(32)
    want = oracle_solve(orc, p, L=2)
    got = capi.solve_problem(p, L=2, nslabs_local=nslabs, devices=(0,))
    assert got["status"] == 0 and got["fused"]
    assert got["history"][-1] <= 1e-6 * want["norm_rhs"]
    assert rel_l2(points(got["phi"]), points(want["phi"])) <= 1e-3


@pytest.mark.parametrize("nslabs", [2, 4])
@pytest.mark.parametrize("L,flags,path", [(1, 0, 2), (2, 0, 2), (1, capi.KRY_FORCE_GENERIC, 0),
                                          (2, capi.KRY_FORCE_GENERIC, 0), (3, 0, 2),
                                          (3, capi.KRY_FORCE_GENERIC, 0)])
@pytest.mark.parametrize("bc", [(0, 0, 1, 1, 0, 1), (1, 1, 0, 1, 1, 1)])
def test_mixed_faces_on_slabs(orc, bc, nslabs, L, flags, path):
    """Extrapolating faces across slabs: the fused general-boundary kernels and the
    one-kernel-per-primitive path."""
    p = P.smooth_var(24, bc=bc)
    p.rhs = np.random.default_rng(9).uniform(-1, 1, p.rhs.shape)
    want = oracle_solve(orc, p, L=L, eps=1e-30, max_outer=3)
    got = capi.solve_problem(p, L=L, eps=1e-30, max_outer=3, nslabs_local=nslabs, devices=(0,), flags=flags)
    assert got["path"] == path
    assert got["iters"] == 3
    assert np.max(np.abs(got["history"] - want["history"])) <= 1e-10 * want["norm_rhs"]
    assert rel_l2(points(got["phi"]), points(want["phi"])) <= 1e-10


@pytest.mark.parametrize("L", [1, 2])
@pytest.mark.parametrize("nslabs", [2, 4])
@pytest.mark.parametrize("overlap", ["0", "1"])
def test_split_launch_with_overlapped_exchange(orc, monkeypatch, L, overlap, nslabs):
    """Short chunks force the edge/interior split of the stencil kernels, so the
    halo exchange runs beside the interior chunks; results must not change."""
    monkeypatch.setenv("KRY_ZC", "4")
    monkeypatch.setenv("KRY_OVERLAP", overlap)
    p = P.smooth_var(48)
    p.rhs = np.random.default_rng(8).uniform(-1, 1, p.rhs.shape)
    one = capi.solve_problem(p, L=L, eps=1e-30, max_outer=6)
    many = capi.solve_problem(p, L=L, eps=1e-30, max_outer=6, nslabs_local=nslabs, devices=(0,))
    assert one["fused"] and many["fused"] and many["iters"] == 6
    assert np.max(np.abs(one["history"] - many["history"])) <= 1e-10 * one["norm_rhs"]
    assert rel_l2(points(many["phi"]), points(one["phi"])) <= 1e-10


@pytest.mark.parametrize("L", [1, 2])
def test_mixed_faces_on_slabs_longer_run(orc, L):
    """Twelve passes on 1, 2 and 4 slabs stay together to rounding noise."""
    p = P.smooth_var((40, 24, 32), bc=(0, 1, 1, 0, 1, 1))
    p.rhs = np.random.default_rng(10).uniform(-1, 1, p.rhs.shape)
    runs = [capi.solve_problem(p, L=L, eps=1e-30, max_outer=12, nslabs_local=ns, devices=(0,)) for ns in (1, 2, 4)]
    assert all(r["path"] == 2 and r["iters"] == 12 for r in runs)
    for r in runs[1:]:
        assert np.max(np.abs(r["history"][:6] - runs[0]["history"][:6])) <= 1e-9 * runs[0]["norm_rhs"]


def test_repeated_multi_slab_solves_are_reproducible(orc):
    """Small grids make the kernels short and the cross-slab ordering tight: the
    same solve, repeated, must return the same history every time."""
    p = P.reference_test1(32)
    p.rhs = P.interior_only(np.random.default_rng(5).uniform(-1, 1, p.rhs.shape))
    ref = None
    for trial in range(40):
        for ns in (4, 8):
            r = capi.solve_problem(p, L=2, eps=1e-30, max_outer=5, nslabs_local=ns, devices=(0,))
            if ref is None:
                ref = {4: None, 8: None}
            if ref[ns] is None:
                ref[ns] = r["history"]
            assert np.array_equal(r["history"], ref[ns]), (trial, ns)
    assert np.max(np.abs(ref[4] - ref[8])) <= 1e-10 * ref[4][0]


def test_sixteen_slabs_on_one_device_with_overlapped_exchange(orc, monkeypatch):
    """The largest slab count a process may own (KRY_MAX_SLABS = 16) on ONE device: 32 streams on
    the device's four hardware queues, every stencil kernel launched in two parts with the
    exchange of 15 interfaces on the second streams in between.  Every event is recorded (host
    order) before anything waits for it, so no hardware queue can hold a wait in front of its own
    producer; this run is the check of that argument at the largest supported size."""
    monkeypatch.setenv("KRY_ZC", "2")
    monkeypatch.setenv("KRY_OVERLAP", "1")
    p = P.smooth_var((24, 20, 128))
    p.rhs = np.random.default_rng(12).uniform(-1, 1, p.rhs.shape)
    one = capi.solve_problem(p, L=2, eps=1e-30, max_outer=8)
    runs = [capi.solve_problem(p, L=2, eps=1e-30, max_outer=8, nslabs_local=16, devices=(0,)) for _ in range(3)]
    for r in runs:
        assert r["fused"] and r["iters"] == 8
        assert np.array_equal(r["history"], runs[0]["history"])
        assert np.max(np.abs(r["history"] - one["history"])) <= 1e-10 * one["norm_rhs"]
        assert rel_l2(points(r["phi"]), points(one["phi"])) <= 1e-10


LINK_WORKER = r"""
import sys, json
import numpy as np
sys.path.insert(0, %(root)r)
from krylov_amd import capi, problems as P
kind, L, flags, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
if kind == "const":
    p = P.reference_test1(32)
    p.rhs = P.interior_only(np.random.default_rng(5).uniform(-1, 1, p.rhs.shape))
else:
    p = P.smooth_var((40, 24, 32), bc=(0, 1, 1, 0, 1, 1)) if kind == "mixed" else P.smooth_var(32)
    p.rhs = np.random.default_rng(6).uniform(-1, 1, p.rhs.shape)
r = capi.solve_problem(p, L=L, eps=1e-30, max_outer=6, nslabs_local=2, devices=(0,), flags=flags)
np.savez(out, phi=r["phi"], history=r["history"], iters=r["iters"], status=r["status"], slab_link=r["slab_link"])
print(r["slab_link_note"])
"""


@pytest.mark.parametrize("overlap", ["0", "1"])
@pytest.mark.parametrize("kind,L,flags", [("var", 1, 0), ("var", 2, 0), ("var", 3, 0), ("const", 2, 0),
                                          ("mixed", 2, 0), ("mixed", 1, 0), ("var", 2, capi.KRY_FORCE_GENERIC)])
def test_slabs_linked_through_device_memory_reproduce_the_copies(tmp_path, kind, L, flags, overlap):
    """Two slabs of one process linked through each other's device memory (mailbox all-reduce by the
    reducing kernels, ghost planes pushed by the sender's kernel) against the same slabs exchanging by
    copies and events: same history, same solution, bit for bit (the sums are added in slab order on
    both routes).  Here both slabs share the one device, which KRY_SLAB_LINK=1 permits; with a device
    per slab the link is the default.  Linked kernels of the two slabs wait for each other, so their
    four streams must sit on four different hardware queues of the device; a process hands its queues
    out in turn, which a fresh process makes certain -- hence one process per solve here.  (Where the
    streams do share a queue the trial exchange at creation times out and the copies are used.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runs = {}
    for link in ("0", "1"):
        out = str(tmp_path / ("link%s.npz" % link))
        env = dict(os.environ, KRY_ZC="4", KRY_OVERLAP=overlap, KRY_MAIL_TIMEOUT_MS="5000", KRY_SLAB_LINK=link)
        res = subprocess.run([sys.executable, "-c", LINK_WORKER % {"root": root}, kind, str(L), str(flags), out], env=env,
                             capture_output=True, text=True, timeout=110)
        assert res.returncode == 0, res.stderr[-2000:]
        runs[link] = (np.load(out), res.stdout.strip().splitlines()[-1])
    (a, note_a), (b, note_b) = runs["0"], runs["1"]
    assert int(a["slab_link"]) == 0 and note_a == "KRY_SLAB_LINK=0"
    if int(b["slab_link"]) == 0 and note_b == "the trial exchange failed":
        # the two slabs' streams share a hardware queue on this box: the library noticed and kept the copies
        # (by design); the solve below must then equal the copies trivially, and the link itself is not exercised
        assert np.array_equal(a["history"], b["history"]) and np.array_equal(a["phi"], b["phi"])
        pytest.skip("linked kernels of two slabs cannot run side by side on this device (streams share a hardware queue)")
    assert int(b["slab_link"]) == 1 and note_b == "on", note_b
    assert int(a["iters"]) == int(b["iters"]) == 6 and int(a["status"]) == int(b["status"])
    assert np.array_equal(a["history"], b["history"])
    assert np.array_equal(a["phi"], b["phi"])


def test_slab_link_is_not_made_where_kernels_could_not_run_side_by_side(orc, monkeypatch):
    """Linked slabs wait for each other inside kernels.  Two slabs on one device are linked only on
    request, more than two never (their streams would share hardware queues): the copies stay."""
    p = P.smooth_var(32)
    monkeypatch.delenv("KRY_SLAB_LINK", raising=False)
    r = capi.solve_problem(p, L=2, eps=1e-30, max_outer=2, nslabs_local=2, devices=(0,))
    assert r["slab_link"] == 0 and "share a device" in r["slab_link_note"]
    monkeypatch.setenv("KRY_SLAB_LINK", "1")
    r = capi.solve_problem(p, L=2, eps=1e-30, max_outer=2, nslabs_local=4, devices=(0,))
    assert r["slab_link"] == 0 and "share a device" in r["slab_link_note"] and r["iters"] == 2
    r = capi.solve_problem(p, L=2, eps=1e-30, max_outer=2)
    assert r["slab_link"] == 0 and r["slab_link_note"] == "one slab"
