"""kry_op_dot_static: the weighted dot product / norm for callers that cannot
allocate, against kry_op_dot / kry_op_l2norm (same kernel, same summation order:
equal bit for bit) and against the CPU oracle's Krylov::dot_product / get_l2Norm.
Two calls in a row share the GPU's static scratch array; the second one works on a
smaller field.  Also: a field with more row groups than the kernel's grid cap, and two
calls recorded by a stream capture and replayed as a graph."""
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


def _oracle_value(orc, mx, a, b, norm_mode):
    cfg = orc.make_cfg(mx, (1.0 / mx[0], 1.0 / mx[1], 1.0 / mx[2]))
    return orc.l2norm(cfg, a) if norm_mode else orc.dot(cfg, a, b)


def test_static_dot_on_a_field_larger_than_its_grid(orc):
    """64 x 130 x 130 points: 17161 grid rows, 4291 row groups, more than the 4096 blocks the kernel is
    launched with at most -- every partial-sum slot of the scratch array is in use and blocks walk several
    row groups.  Checked against the oracle's sums (reference order: 1e-12) and against kry_op_dot (bit for bit)."""
    L = capi.lib()
    dev = Dev()
    mx = (63, 129, 129)
    try:
        res = dev.put(np.zeros(1))
        for norm_mode in (0, 1):
            a, b = fields(mx, 5 + norm_mode)
            if norm_mode:
                b = a
            da = dev.put(a)
            db = da if norm_mode else dev.put(b)
            assert L.kry_op_dot_static(0, da, db, (C.c_int * 3)(*mx), norm_mode, res, None) == 0, L.kry_last_error()
            got = float(dev.get(res)[0])
            want = _oracle_value(orc, mx, a, b, norm_mode)
            assert abs(got - want) <= 1e-12 * abs(want)
            with capi.Solver(mx, (1.0 / mx[0], 1.0 / mx[1], 1.0 / mx[2]), L=1) as s:
                assert got == (s.op_l2norm(a) if norm_mode else s.op_dot(a, b))
    finally:
        dev.close()


def test_static_dot_recorded_by_a_stream_capture(orc):
    """What the entry point is for: after kry_device_check (which asks the runtime for the scratch array's
    address), two calls are recorded by a stream capture -- nothing runs, nothing is allocated -- and the
    graph is launched twice; both results equal the oracle's sums and those of plain calls."""
    L = capi.lib()
    vp = C.c_void_p
    L.hipStreamCreate.argtypes = [C.POINTER(vp)]
    L.hipStreamDestroy.argtypes = [vp]
    L.hipStreamSynchronize.argtypes = [vp]
    L.hipStreamBeginCapture.argtypes = [vp, C.c_int]
    L.hipStreamEndCapture.argtypes = [vp, C.POINTER(vp)]
    L.hipGraphInstantiate.argtypes = [C.POINTER(vp), vp, vp, vp, C.c_size_t]
    L.hipGraphLaunch.argtypes = [vp, vp]
    L.hipGraphExecDestroy.argtypes = [vp]
    L.hipGraphDestroy.argtypes = [vp]
    L.hipMemsetAsync.argtypes = [vp, C.c_int, C.c_size_t, vp]
    capi.device_check(0)
    dev = Dev()
    stream, graph, gexec = vp(), vp(), vp()
    try:
        assert L.hipStreamCreate(C.byref(stream)) == 0
        cases = [((24, 20, 18), 0, 11), ((12, 40, 9), 1, 12)]
        res = dev.put(np.zeros(2))
        want, plain, args = [], [], []
        for mx, norm_mode, seed in cases:
            a, b = fields(mx, seed)
            if norm_mode:
                b = a
            want.append(_oracle_value(orc, mx, a, b, norm_mode))
            da = dev.put(a)
            args.append((da, da if norm_mode else dev.put(b), mx, norm_mode))
        for k, (da, db, mx, norm_mode) in enumerate(args):
            assert L.kry_op_dot_static(0, da, db, (C.c_int * 3)(*mx), norm_mode, vp(res.value + 8 * k), stream) == 0
        assert L.hipStreamSynchronize(stream) == 0
        plain = list(dev.get(res, 2))
        assert L.hipStreamBeginCapture(stream, 0) == 0
        for k, (da, db, mx, norm_mode) in enumerate(args):
            st = L.kry_op_dot_static(0, da, db, (C.c_int * 3)(*mx), norm_mode, vp(res.value + 8 * k), stream)
            assert st == 0, L.kry_last_error()
        assert L.hipStreamEndCapture(stream, C.byref(graph)) == 0 and graph.value
        assert L.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0) == 0
        for _ in range(2):
            assert L.hipMemsetAsync(res, 0, 16, stream) == 0          # the graph, not the earlier calls, writes the results
            assert L.hipGraphLaunch(gexec, stream) == 0
            assert L.hipStreamSynchronize(stream) == 0
            got = list(dev.get(res, 2))
            assert got == plain
            assert all(abs(g - w) <= 1e-12 * abs(w) for g, w in zip(got, want))
    finally:
        if gexec.value:
            L.hipGraphExecDestroy(gexec)
        if graph.value:
            L.hipGraphDestroy(graph)
        if stream.value:
            L.hipStreamDestroy(stream)
        dev.close()
