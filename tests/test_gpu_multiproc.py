"""One process per z slab (the layout of `bench.py --gpus N`), all on GPU 0,
with a host-staged gloo transport plugged into the C ABI's callback hooks.
Checks the library's multi-process orchestration -- per-process scalar state
advanced after an inter-process all-reduce, operator outputs exchanged between
processes -- against the single-process solve."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from krylov_amd import problems as P

HERE = os.path.dirname(os.path.abspath(__file__))
gpu = pytest.mark.gpu


def make_problem(kind):
    rng = np.random.default_rng(23)
    if kind.endswith("_generic") and kind != "mixed_generic":
        kind = kind[:-len("_generic")]
    if kind == "const":
        p = P.reference_test1(32)
        p.rhs = P.interior_only(rng.uniform(-1, 1, p.rhs.shape))
    elif kind == "const_faces":
        # constant coefficients on faces that are not all Dirichlet: the operator's input passes through do_BCs
        # (Krylov.C:516), here with ghost cells of neighbouring sub-domains among the cells that treatment reads
        p = P.reference_test1(32)
        p.bc = (1, 0, 1, 1, 0, 1)
        p.rhs = rng.uniform(-1, 1, p.rhs.shape)
        p.phi = 1e-2 * rng.uniform(-1, 1, p.phi.shape)
    elif kind == "var":
        p = P.smooth_var(32)
        p.rhs = rng.uniform(-1, 1, p.rhs.shape)
    elif kind == "offdiag":
        p = P.offdiag_var(32)
        p.rhs = rng.uniform(-1, 1, p.rhs.shape)
    elif kind == "offdiag_edge":
        # Dxy away from zero on extrapolating faces: edge points read edge ghosts
        p = P.offdiag_var(32)
        p.bc = (1, 1, 1, 0, 1, 1)
        p.Dxy = 0.05 + 0.1 * rng.uniform(0, 1, p.Dxy.shape)
        p.rhs = rng.uniform(-1, 1, p.rhs.shape)
    elif kind in ("mixed32",):
        p = P.smooth_var(32, bc=(1, 0, 0, 1, 1, 1))
        p.rhs = rng.uniform(-1, 1, p.rhs.shape)
    elif kind == "wide":
        # 512 points in x: a sub-domain of a split in x still has two 128-point tile columns, one of which reads no
        # exchanged cell (the stencil kernels run it while the ghost cells of their input are on their way)
        p = P.random_fields((512, 24, 32), 29, bc=(0, 1, 0, 0, 1, 0), var_coef=True, use_offdiag=False)
    elif kind in ("mixed", "mixed_generic"):
        p = P.smooth_var(24, bc=(0, 0, 1, 1, 0, 1))
        p.rhs = rng.uniform(-1, 1, p.rhs.shape)
    else:
        raise KeyError(kind)
    return p


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_ranks(world, kind, L, max_outer, eps, outdir, nproc=None):
    port = free_port()
    extra = [",".join(str(v) for v in nproc)] if nproc else []
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "mp_slab_worker.py"), str(r), str(world), str(port),
                               kind, str(L), str(max_outer), repr(eps), outdir] + extra) for r in range(world)]
    try:
        codes = [q.wait(timeout=240) for q in procs]
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    assert codes == [0] * world
    return [np.load(os.path.join(outdir, "r%d.npz" % r)) for r in range(world)]


@gpu
@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("kind,L", [("const", 1), ("var", 2), ("mixed", 2), ("mixed", 1), ("mixed_generic", 2),
                                    ("var", 3), ("mixed", 4), ("offdiag_edge", 2), ("const_faces", 1),
                                    ("const_faces", 2)])
def test_processes_reproduce_the_single_process_solve(tmp_path, kind, L, world):
    from helpers import points, rel_l2
    from krylov_amd import capi
    p = make_problem(kind)
    flags = capi.KRY_FORCE_GENERIC if kind == "mixed_generic" else 0
    one = capi.solve_problem(p, L=L, eps=1e-30, max_outer=3, flags=flags)
    parts = run_ranks(world, kind, L, 3, 1e-30, str(tmp_path))
    mzl = p.mx[2] // world
    for r, d in enumerate(parts):
        assert int(d["iters"]) == 3 and int(d["fused"]) == one["path"]
        assert float(d["norm_rhs"]) == pytest.approx(one["norm_rhs"], rel=1e-13)
        assert np.max(np.abs(d["hist"] - one["history"])) <= 1e-10 * one["norm_rhs"]
        mine = d["phi"][1:mzl + 2, 1:-1, 1:-1]
        ref = one["phi"][r * mzl + 1: r * mzl + mzl + 2, 1:-1, 1:-1]
        assert np.max(np.abs(mine - ref)) <= 1e-10 * np.max(np.abs(ref))
    # every rank holds the same history, bit for bit
    for d in parts[1:]:
        assert np.array_equal(d["hist"], parts[0]["hist"])
    # and it is the history of one process driving the same slabs (the transport adds the
    # per-slab sums in rank order, as the library does for its own slabs)
    same = capi.solve_problem(p, L=L, eps=1e-30, max_outer=3, flags=flags, nslabs_local=world, devices=(0,))
    assert np.array_equal(parts[0]["hist"], same["history"])


ALL_SPLITS = [(2, 1, 1), (1, 2, 1), (2, 2, 1), (1, 2, 2), (2, 2, 2)]
# three kinds on every split, the others on the x-y split and the full split (one of them on the full split only)
CART_CASES = [(k, L, n) for (k, L) in (("var", 2), ("mixed32", 2), ("offdiag", 2)) for n in ALL_SPLITS] + \
             [(k, L, n) for (k, L) in (("const", 2), ("var", 1), ("var_generic", 2), ("offdiag", 3), ("offdiag_edge", 2),
                                       ("offdiag_edge", 1), ("const_faces", 2), ("const_faces", 1)) for n in ((2, 2, 1), (2, 2, 2))] + [("const_generic", 1, (2, 2, 2))]


@gpu
@pytest.mark.parametrize("kind,L,nproc", CART_CASES)
def test_cartesian_split_reproduces_the_single_process_solve(tmp_path, kind, L, nproc):
    """Sub-domains in x and y as well (the reference's mpi_manager_3D layout, up to its
    driver's 2 x 2 x 2): three passes against the undivided solve."""
    from helpers import points
    from krylov_amd import capi
    p = make_problem(kind)
    flags = capi.KRY_FORCE_GENERIC if kind.endswith("_generic") else 0
    one = capi.solve_problem(p, L=L, eps=1e-30, max_outer=3, flags=flags)
    world = nproc[0] * nproc[1] * nproc[2]
    parts = run_ranks(world, kind, L, 3, 1e-30, str(tmp_path), nproc)
    m = [p.mx[d] // nproc[d] for d in range(3)]
    for r, d in enumerate(parts):
        c = (r // (nproc[1] * nproc[2]), (r // nproc[2]) % nproc[1], r % nproc[2])
        assert int(d["iters"]) == 3
        assert int(d["fused"]) == (0 if flags else 2)
        assert float(d["norm_rhs"]) == pytest.approx(one["norm_rhs"], rel=1e-13)
        assert np.max(np.abs(d["hist"] - one["history"])) <= 1e-10 * one["norm_rhs"]
        mine = d["phi"][1:-1, 1:-1, 1:-1]
        ref = one["phi"][c[2] * m[2] + 1: c[2] * m[2] + m[2] + 2, c[1] * m[1] + 1: c[1] * m[1] + m[1] + 2,
                         c[0] * m[0] + 1: c[0] * m[0] + m[0] + 2]
        assert np.max(np.abs(mine - ref)) <= 1e-10 * np.max(np.abs(ref))
    for d in parts[1:]:
        assert np.array_equal(d["hist"], parts[0]["hist"])


@gpu
@pytest.mark.parametrize("world,kind,L,nproc", [(2, "var", 2, None), (4, "var", 2, None), (2, "const", 1, None),
                                                (4, "mixed", 2, None), (2, "mixed_generic", 2, None),
                                                (4, "var", 2, (1, 2, 2)), (4, "var", 1, (2, 2, 1)),
                                                (8, "offdiag", 2, (2, 2, 2)), (2, "var", 3, None),
                                                (4, "mixed", 3, (1, 2, 2)), (2, "wide", 2, (2, 1, 1)),
                                                (4, "wide", 1, (2, 1, 2))])
def test_mapped_memory_exchange_reproduces_the_transport(tmp_path, monkeypatch, world, kind, L, nproc):
    """The dot products all-reduced by the reduction kernels themselves through mailboxes, and the
    ghost planes pushed into landing slots, both in memory the processes map from each other (HIP
    IPC; here all on one device), against the same run through the transport: the sums are added in
    rank order either way and planes are copies, so histories and solutions are equal bit for bit."""
    monkeypatch.setenv("KRY_MAIL_TIMEOUT_MS", "20000")
    (tmp_path / "0").mkdir()
    plain = run_ranks(world, kind, L, 4, 1e-30, str(tmp_path / "0"), nproc)
    for mode in (3,) if L == 3 else (1, 3) if world == 8 else (1, 2, 3):
        (tmp_path / str(mode)).mkdir()
        monkeypatch.setenv("KRY_TEST_MAIL", str(mode))
        mail = run_ranks(world, kind, L, 4, 1e-30, str(tmp_path / str(mode)), nproc)
        for a, b in zip(plain, mail):
            assert int(b["iters"]) == 4 and int(b["status"]) == int(a["status"])
            assert np.array_equal(a["hist"], b["hist"]), mode
            assert np.array_equal(a["phi"], b["phi"]), mode
        if nproc is not None and mode >= 2 and (nproc[0] == 1 or kind == "wide"):   # (a 16-point sub-domain in x has no inner tile)
            # a Cartesian split with the planes travelling through mapped memory: the stencil kernels ran their
            # inner tiles while the ghost cells of their input were on their way (3 of the 4 launches of an l = 2 pass,
            # 1 of 2 for l = 1; the host-staged transport of the plain run cannot overlap)
            assert all(int(a["overlapped"]) == 0 for a in plain)
            assert all(int(b["overlapped"]) >= 4 * (1 if L == 1 else L) for b in mail), [int(b["overlapped"]) for b in mail]


@gpu
@pytest.mark.parametrize("world,kind,nproc", [(4, "var", None), (4, "var", (2, 2, 1))])
def test_mapped_memory_exchange_over_many_passes(tmp_path, monkeypatch, world, kind, nproc):
    """150 passes (900 all-reduces, 600 and more plane exchanges per rank, slots and sequence words
    reused throughout): still the transport's history, bit for bit."""
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    plain = run_ranks(world, kind, 2, 150, 1e-30, str(tmp_path / "a"), nproc)
    monkeypatch.setenv("KRY_TEST_MAIL", "3")
    mail = run_ranks(world, kind, 2, 150, 1e-30, str(tmp_path / "b"), nproc)
    for a, b in zip(plain, mail):
        assert int(a["iters"]) == int(b["iters"]) and int(b["iters"]) > 20
        assert np.array_equal(a["hist"], b["hist"]) and np.array_equal(a["phi"], b["phi"])


@gpu
@pytest.mark.parametrize("world,kind,nproc", [(4, "var", None), (4, "mixed32", (2, 2, 1))])
def test_mapped_memory_alone_needs_no_transport(tmp_path, monkeypatch, world, kind, nproc):
    """No communicator, no callbacks: connected mailboxes and landing slots carry everything."""
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    plain = run_ranks(world, kind, 2, 4, 1e-30, str(tmp_path / "a"), nproc)
    monkeypatch.setenv("KRY_TEST_MAIL", "3")
    monkeypatch.setenv("KRY_TEST_NO_TRANSPORT", "1")
    alone = run_ranks(world, kind, 2, 4, 1e-30, str(tmp_path / "b"), nproc)
    for a, b in zip(plain, alone):
        assert int(b["iters"]) == 4
        assert np.array_equal(a["hist"], b["hist"]) and np.array_equal(a["phi"], b["phi"])


@gpu
def test_mailbox_allreduce_gives_up_on_an_absent_rank(tmp_path, monkeypatch):
    """A rank whose peer never reaches the dot product stops after KRY_MAIL_TIMEOUT_MS with a
    communication error instead of waiting for ever."""
    import time
    monkeypatch.setenv("KRY_TEST_MAIL", "1")
    monkeypatch.setenv("KRY_TEST_MAIL_ABSENT", "1")
    monkeypatch.setenv("KRY_MAIL_TIMEOUT_MS", "300")
    port = free_port()
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "mp_slab_worker.py"), str(r), "2", str(port), "var", "2",
                               "3", "1e-30", str(tmp_path)]) for r in range(2)]
    try:
        codes = [q.wait(timeout=120) for q in procs]
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    assert codes == [0, 0]
    msg = (tmp_path / "r0.err").read_text()
    assert "communication error" in msg, msg
    assert time.time() - t0 < 60


@gpu
def test_mapped_memory_planes_give_up_once(tmp_path, monkeypatch):
    """Ghost planes through mapped memory, the neighbour gone after the load: the first exchange of
    the batch of passes waits KRY_MAIL_TIMEOUT_MS and fails the solve, and the dozens of exchanges
    queued behind it return at once instead of waiting that long each."""
    monkeypatch.setenv("KRY_TEST_MAIL", "3")
    monkeypatch.setenv("KRY_TEST_NO_TRANSPORT", "1")
    monkeypatch.setenv("KRY_TEST_MAIL_ABSENT", "2")
    monkeypatch.setenv("KRY_MAIL_TIMEOUT_MS", "500")
    port = free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "mp_slab_worker.py"), str(r), "2", str(port), "var", "2",
                               "0", "1e-30", str(tmp_path)]) for r in range(2)]
    try:
        codes = [q.wait(timeout=120) for q in procs]
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    assert codes == [0, 0]
    msg = (tmp_path / "r0.err").read_text()
    assert "communication error" in msg, msg
    assert float(msg.split(" s:")[0]) < 3.0, msg       # 8 passes x 4 exchanges would take 16 s


@gpu
def test_processes_full_solve(tmp_path):
    from krylov_amd import capi
    p = make_problem("var")
    one = capi.solve_problem(p, L=2, eps=1e-6)
    parts = run_ranks(2, "var", 2, 0, 1e-6, str(tmp_path))
    for d in parts:
        assert int(d["status"]) == 0
        assert d["hist"][-1] <= 1e-6 * one["norm_rhs"]
        assert abs(int(d["iters"]) - one["iters"]) <= max(8, one["iters"] // 3)


@gpu
def test_rccl_transport_selftest():
    """RCCL itself on this device: one-rank communicator, all-reduce, grouped
    send/receive to self, data verified inside the library."""
    from krylov_amd import capi
    assert capi.comm_selftest(0) == 0


def _gpu_count():
    import ctypes as C
    from krylov_amd import capi
    n = C.c_int(0)
    L = capi.lib()
    L.hipGetDeviceCount.argtypes = [C.POINTER(C.c_int)]
    return n.value if L.hipGetDeviceCount(C.byref(n)) == 0 else 0


@gpu
@pytest.mark.parametrize("kind,L,nproc", [("var", 2, None), ("mixed", 2, None), ("var", 2, (1, 2, 1))])
def test_rccl_transport_between_gpus(tmp_path, monkeypatch, kind, L, nproc):
    """One process per GPU with RCCL as the transport (the layout of `bench.py --gpus N`).
    Needs at least two GPUs; the per-rank sums are all-reduced by RCCL, whose summation
    order is its own, so the comparison is to rounding noise, not bit for bit."""
    from krylov_amd import capi
    if _gpu_count() < 2:
        pytest.skip("needs two GPUs")
    monkeypatch.setenv("KRY_TEST_RCCL", "1")
    p = make_problem(kind)
    one = capi.solve_problem(p, L=L, eps=1e-30, max_outer=3)
    parts = run_ranks(2, kind, L, 3, 1e-30, str(tmp_path), nproc)
    for d in parts:
        assert int(d["iters"]) == 3
        assert np.max(np.abs(d["hist"] - one["history"])) <= 1e-10 * one["norm_rhs"]
