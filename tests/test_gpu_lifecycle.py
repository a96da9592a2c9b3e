"""Handles give back what they take: device memory after many create / solve / destroy
cycles on every kernel path, slab count and communication set-up equals the memory before."""
import ctypes as C

import numpy as np
import pytest

from krylov_amd import capi, problems as P

pytestmark = pytest.mark.gpu


def free_bytes():
    L = capi.lib()
    L.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.hipDeviceSynchronize.argtypes = []
    assert L.hipDeviceSynchronize() == 0
    f, t = C.c_size_t(), C.c_size_t()
    assert L.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value


def cycle():
    for p, L, flags, nslabs in ((P.reference_test1(32), 2, 0, 1), (P.smooth_var(32), 1, 0, 2),
                                (P.smooth_var(24, bc=(0, 0, 1, 1, 0, 1)), 2, 0, 1), (P.offdiag_var(24), 3, 0, 4),
                                (P.smooth_var(24, bc=(1,) * 6), 2, capi.KRY_FORCE_GENERIC, 1)):
        r = capi.solve_problem(p, L=L, eps=1e-6, flags=flags, nslabs_local=nslabs, devices=(0,))
        assert r["status"] == 0
    p = P.smooth_var(32)
    with capi.Solver(p.mx, p.delx, p.bc, L=2, eps=0.0) as s:          # resident stepping, updates, a re-solve
        s.load_var(p.phi, p.rhs, p.lam, p.Dxx, p.Dyy, p.Dzz, None, False)
        s.iterate(3)
        s.update_rhs(p.rhs * 2.0)
        s.update_phi(p.phi)
        s.reset()
        s.iterate(2)
        s.op_l2norm(p.rhs)


def test_device_memory_returns_after_many_handles():
    cycle()                                   # first use: runtime pools, code objects
    before = free_bytes()
    for _ in range(10):
        cycle()
    after = free_bytes()
    assert abs(after - before) <= 8 << 20, (before, after)     # (the runtime's own pools may move by a few MiB)


def test_failed_create_leaves_nothing_behind():
    before = free_bytes()
    for _ in range(20):
        with pytest.raises(Exception):
            capi.Solver((32, 32, 32), (1 / 32.,) * 3, (0,) * 6, L=99)          # L outside 1..8
        with pytest.raises(Exception):
            capi.Solver((32, 32, 30), (1 / 32.,) * 3, (0,) * 6, L=2, nslabs=4, slab_begin=0, slab_count=4,
                        devices=(0, 0, 0, 0))                                   # 30 planes do not divide into 4 slabs
    assert abs(free_bytes() - before) <= 8 << 20


def test_independent_handles_on_concurrent_host_threads():
    """The reference's solver objects are independent of each other; so are handles: four host
    threads, each with a handle of its own on the same GPU, obtain the bits of a serial run."""
    import threading
    probs = [(P.reference_test1(32), 2), (P.smooth_var(32), 1), (P.smooth_var(24, bc=(0, 0, 1, 1, 0, 1)), 2),
             (P.offdiag_var(24), 3)]
    serial = [capi.solve_problem(p, L=L, eps=1e-8) for p, L in probs]
    out = [None] * len(probs)

    def work(k):
        p, L = probs[k]
        for _ in range(3):
            out[k] = capi.solve_problem(p, L=L, eps=1e-8)

    th = [threading.Thread(target=work, args=(k,)) for k in range(len(probs))]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=200)
    for a, b in zip(serial, out):
        assert b is not None and b["status"] == a["status"] == 0
        assert np.array_equal(a["history"], b["history"]) and np.array_equal(a["phi"], b["phi"])
