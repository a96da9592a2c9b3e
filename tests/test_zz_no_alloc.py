"""kry_op_dot_static: the weighted dot product / norm for callers that cannot
allocate, against kry_op_dot / kry_op_l2norm (same kernel, same summation order:
equal bit for bit).  Two calls in a row share the GPU's static scratch array; the
second one works on a smaller field."""
import ctypes as C

import numpy as np
import pytest

from krylov_amd import capi

pytestmark = pytest.mark.gpu


class Dev:
    """Plain device buffers through the HIP runtime the solver library is linked to."""

    def __init__(self):
        self.L = capi.lib()
        self.L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.L.hipFree.argtypes = [C.c_void_p]
        self.L.hipDeviceSynchronize.argtypes = []
        self.bufs = []

    def put(self, a):
        d = C.c_void_p()
        assert self.L.hipMalloc(C.byref(d), a.nbytes) == 0
        assert self.L.hipMemcpy(d, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0
        self.bufs.append(d)
        return d

    def get(self, d, n=1):
        out = np.empty(n)
        assert self.L.hipDeviceSynchronize() == 0
        assert self.L.hipMemcpy(out.ctypes.data_as(C.c_void_p), d, out.nbytes, 2) == 0
        return out

    def close(self):
        for d in self.bufs:
            self.L.hipFree(d)
        self.bufs = []


def fields(mx, seed):
    rng = np.random.default_rng(seed)
    shape = (mx[2] + 3, mx[1] + 3, mx[0] + 3)
    return rng.uniform(-1, 1, shape), rng.uniform(-1, 1, shape)


def test_static_dot_equals_op_dot_on_two_calls_in_a_row():
    L = capi.lib()
    dev = Dev()
    try:
        res = dev.put(np.full(2, -7.0))
        # first call: a dot product on a larger field; second: a norm on a smaller one
        cases = [((40, 24, 20), 0, 1), ((16, 12, 8), 1, 2)]
        want, args = [], []
        for mx, norm_mode, seed in cases:
            a, b = fields(mx, seed)
            if norm_mode:
                b = a
            with capi.Solver(mx, (1.0 / mx[0], 1.0 / mx[1], 1.0 / mx[2]), L=1) as s:
                want.append(s.op_l2norm(a) if norm_mode else s.op_dot(a, b))
            args.append((dev.put(a), dev.put(b) if not norm_mode else None, mx, norm_mode))
        got = []
        for da, db, mx, norm_mode in args:           # back to back, nothing in between
            st = L.kry_op_dot_static(0, da, db if db is not None else da, (C.c_int * 3)(*mx), norm_mode, res, None)
            assert st == 0, L.kry_last_error()
            got.append(float(dev.get(res)[0]))
        assert got[0] == want[0]
        assert got[1] == want[1]
        # and once more in the opposite order: what the previous call left in the scratch does not matter
        for k in (1, 0):
            da, db, mx, norm_mode = args[k]
            st = L.kry_op_dot_static(0, da, db if db is not None else da, (C.c_int * 3)(*mx), norm_mode, res, None)
            assert st == 0, L.kry_last_error()
            assert float(dev.get(res)[0]) == want[k]
    finally:
        dev.close()


def test_static_dot_refuses_oversized_fields_and_null_pointers():
    L = capi.lib()
    dev = Dev()
    try:
        d = dev.put(np.zeros(8))
        st = L.kry_op_dot_static(0, d, d, (C.c_int * 3)(4000, 4000, 4000), 0, d, None)
        assert st == -1 and b"exceeds the limit" in L.kry_last_error()
        assert L.kry_op_dot_static(0, None, d, (C.c_int * 3)(8, 8, 8), 0, d, None) == -1
        assert L.kry_op_dot_static(0, d, d, (C.c_int * 3)(8, 0, 8), 0, d, None) == -1
    finally:
        dev.close()
