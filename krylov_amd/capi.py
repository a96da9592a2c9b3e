"""ctypes binding of libkrylov_amd.so (the C ABI in include/krylov_amd.h).

Used by the tests and by bench.py; the C++ adaptor classes under
krylov_amd/compat call the same entry points directly.  There is no CPU
fallback: without the built library, or without a HIP device, calls fail.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# KRY_LIB_PATH: another build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("KRY_LIB_PATH") or os.path.join(_HERE, "libkrylov_amd.so")

KRY_MAX_SLABS = 16
KRY_UID_BYTES = 128
KRY_MAIL_HANDLE_BYTES = 64
KRY_MAIL_SUMS, KRY_MAIL_PLANES = 1, 2
KRY_FORCE_GENERIC = 1

STATUS = {0: "ok", 1: "alpha is nan", 3: "norm is nan", 4: "iteration cap reached",
          -1: "invalid argument", -2: "HIP error", -3: "communication error", -4: "invalid state"}

# every symbol include/krylov_amd.h declares
SYMBOLS = [
    "kry_create", "kry_destroy", "kry_comm_unique_id", "kry_comm_init", "kry_comm_init_callbacks", "kry_comm_init_host_callbacks", "kry_device_check",
    "kry_comm_selftest", "kry_mail_export", "kry_mail_connect", "kry_mail_enable", "kry_solve_const",
    "kry_solve_var", "kry_load_const", "kry_load_var", "kry_iterate", "kry_fetch_phi", "kry_fetch_residual",
    "kry_update_rhs", "kry_update_phi", "kry_update_lambda", "kry_reset", "kry_resolve",
    "kry_set_profile", "kry_get_stats", "kry_slab_link_note", "kry_get_history", "kry_get_step_times", "kry_op_do_bcs", "kry_op_spmv_const", "kry_op_spmv_var",
    "kry_op_dot", "kry_op_l2norm", "kry_op_dot_static", "kry_status_string", "kry_last_error",
]


class KryDesc(C.Structure):
    _fields_ = [
        ("mx", C.c_int * 3),
        ("delx", C.c_double * 3),
        ("bc_type", C.c_int * 6),
        ("L", C.c_int),
        ("eps", C.c_double),
        ("max_outer", C.c_int),
        ("flags", C.c_int),
        ("nslabs", C.c_int),
        ("slab_begin", C.c_int),
        ("slab_count", C.c_int),
        ("devices", C.c_int * KRY_MAX_SLABS),
        ("nproc", C.c_int * 3),
        ("coord", C.c_int * 3),
    ]


class KryStats(C.Structure):
    _fields_ = [
        ("status", C.c_int),
        ("iterations", C.c_int),
        ("fused", C.c_int),
        ("norm_rhs", C.c_double),
        ("norm_init", C.c_double),
        ("norm_final", C.c_double),
        ("solve_ms", C.c_double),
        ("stage_ms", C.c_double),
        ("bytes_per_iter", C.c_double),
        ("kernel_ms", C.c_double * 8),
        ("kernel_calls", C.c_int * 8),
        ("tile_rows", C.c_int),
        ("chunk_planes", C.c_int),
        ("stream_blocks", C.c_int),
        ("enqueue_ms", C.c_double),
        ("comm_wait_ms", C.c_double * 2),
        ("comm_wait_calls", C.c_int * 2),
        ("rccl_ranks", C.c_int),
        ("staged_fields", C.c_int),
        ("overlapped_exchanges", C.c_int),
        ("slab_link", C.c_int),
    ]


ALLREDUCE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int)
HALO_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                      C.c_ulong)


class KryError(RuntimeError):
    pass


_lib = None
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def lib():
    """Load the shared library.  Loading runs the library's constructor, which
    initialises its HIP runtime (a device count query): the first copy of the runtime
    that touches the GPU in a process must be this one."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KryError("libkrylov_amd.so is not built: run `python -c 'import __graft_entry__ as g; "
                       "g.build()'` or `make -C krylov_amd/csrc`")
    L = C.CDLL(LIB_PATH)
    P, vp, ci, cd = C.POINTER, C.c_void_p, C.c_int, C.c_double
    L.kry_create.restype = vp
    L.kry_create.argtypes = [P(KryDesc), P(ci)]
    L.kry_destroy.restype = None
    L.kry_destroy.argtypes = [vp]
    L.kry_comm_unique_id.restype = ci
    L.kry_comm_unique_id.argtypes = [vp]
    L.kry_comm_init.restype = ci
    L.kry_comm_init.argtypes = [vp, vp, ci, ci]
    L.kry_comm_selftest.restype = ci
    L.kry_comm_selftest.argtypes = [ci]
    L.kry_comm_init_callbacks.restype = ci
    L.kry_comm_init_callbacks.argtypes = [vp, ci, ci, ALLREDUCE_CB, HALO_CB, vp]
    L.kry_comm_init_host_callbacks.restype = ci
    L.kry_comm_init_host_callbacks.argtypes = [vp, ci, ci, ALLREDUCE_CB, HALO_CB, vp]
    L.kry_device_check.restype = ci
    L.kry_device_check.argtypes = [ci]
    L.kry_get_step_times.restype = ci
    L.kry_get_step_times.argtypes = [vp, vp, ci]
    L.kry_mail_export.restype = ci
    L.kry_mail_export.argtypes = [vp, vp]
    L.kry_mail_connect.restype = ci
    L.kry_mail_connect.argtypes = [vp, ci, ci, vp]
    L.kry_mail_enable.restype = ci
    L.kry_mail_enable.argtypes = [vp, ci]
    L.kry_solve_const.restype = ci
    L.kry_solve_const.argtypes = [vp, _dp, _dp, _dp, cd, cd, cd, ci]
    L.kry_solve_var.restype = ci
    L.kry_solve_var.argtypes = [vp, _dp, _dp, _dp, _dp, _dp, _dp, vp, ci, ci]
    L.kry_load_const.restype = ci
    L.kry_load_const.argtypes = [vp, _dp, _dp, _dp, cd, cd, cd]
    L.kry_load_var.restype = ci
    L.kry_load_var.argtypes = [vp, _dp, _dp, _dp, _dp, _dp, _dp, vp, ci]
    for name in ("kry_update_rhs", "kry_update_phi", "kry_update_lambda"):
        getattr(L, name).restype = ci
        getattr(L, name).argtypes = [vp, _dp]
    L.kry_reset.restype = ci
    L.kry_reset.argtypes = [vp]
    L.kry_resolve.restype = ci
    L.kry_resolve.argtypes = [vp, vp, ci]
    L.kry_iterate.restype = ci
    L.kry_iterate.argtypes = [vp, ci]
    L.kry_fetch_phi.restype = ci
    L.kry_fetch_phi.argtypes = [vp, _dp]
    L.kry_fetch_residual.restype = ci
    L.kry_fetch_residual.argtypes = [vp, _dp]
    L.kry_set_profile.restype = ci
    L.kry_set_profile.argtypes = [vp, ci]
    L.kry_get_stats.restype = ci
    L.kry_get_stats.argtypes = [vp, P(KryStats)]
    L.kry_get_history.restype = ci
    L.kry_get_history.argtypes = [vp, vp, ci]
    L.kry_op_do_bcs.restype = ci
    L.kry_op_do_bcs.argtypes = [vp, _dp]
    L.kry_op_spmv_const.restype = ci
    L.kry_op_spmv_const.argtypes = [vp, _dp, _dp, cd, cd, cd, _dp, ci, ci]
    L.kry_op_spmv_var.restype = ci
    L.kry_op_spmv_var.argtypes = [vp, _dp, _dp, _dp, _dp, _dp, vp, ci, _dp, ci, ci]
    L.kry_op_dot.restype = ci
    L.kry_op_dot.argtypes = [vp, _dp, _dp, P(cd)]
    L.kry_op_l2norm.restype = ci
    L.kry_op_l2norm.argtypes = [vp, _dp, P(cd)]
    L.kry_op_dot_static.restype = ci
    L.kry_op_dot_static.argtypes = [ci, vp, vp, C.c_int * 3, ci, vp, vp]
    L.kry_slab_link_note.restype = C.c_char_p
    L.kry_slab_link_note.argtypes = [vp]
    L.kry_status_string.restype = C.c_char_p
    L.kry_status_string.argtypes = [ci]
    L.kry_last_error.restype = C.c_char_p
    L.kry_last_error.argtypes = []
    _lib = L
    return L


def _check(st, what):
    if st < 0:
        raise KryError("%s failed: %s (%s)" % (what, STATUS.get(st, st),
                                                lib().kry_last_error().decode(errors="replace")))
    return st


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def device_check(device=0):
    """Raises KryError unless HIP device `device` accepts work from this library."""
    return _check(lib().kry_device_check(int(device)), "kry_device_check")


def device_count():
    """HIP devices visible to the solver library's runtime."""
    n = C.c_int(0)
    L = lib()
    L.hipGetDeviceCount.argtypes = [C.POINTER(C.c_int)]
    L.hipGetDeviceCount.restype = C.c_int
    return n.value if L.hipGetDeviceCount(C.byref(n)) == 0 else 0


def comm_selftest(device=0):
    return _check(lib().kry_comm_selftest(int(device)), "kry_comm_selftest")


def unique_id():
    buf = C.create_string_buffer(KRY_UID_BYTES)
    _check(lib().kry_comm_unique_id(buf), "kry_comm_unique_id")
    return buf.raw


class Solver:
    """One kry_handle.  `mx` is the extent of the host fields this process
    passes (its own slabs only)."""

    def __init__(self, mx, delx, bc=(0,) * 6, L=2, eps=1e-6, max_outer=0, flags=0,
                 nslabs=1, slab_begin=0, slab_count=1, devices=(0,), nproc=None, coord=None):
        d = KryDesc()
        for k in range(3):
            d.mx[k] = int(mx[k])
            d.delx[k] = float(delx[k])
        for f in range(6):
            d.bc_type[f] = int(bc[f])
        d.L, d.eps, d.max_outer, d.flags = int(L), float(eps), int(max_outer), int(flags)
        d.nslabs, d.slab_begin, d.slab_count = int(nslabs), int(slab_begin), int(slab_count)
        devs = list(devices) + [devices[-1]] * (slab_count - len(devices))
        for i in range(slab_count):
            d.devices[i] = int(devs[i])
        if nproc is not None:            # Cartesian split: this process holds sub-domain `coord`
            for k in range(3):
                d.nproc[k], d.coord[k] = int(nproc[k]), int(coord[k])
            d.nslabs, d.slab_begin = int(nproc[2]), int(coord[2])
        st = C.c_int(0)
        self._h = lib().kry_create(C.byref(d), C.byref(st))
        if not self._h:
            raise KryError("kry_create failed: %s (%s)" % (
                STATUS.get(st.value, st.value), lib().kry_last_error().decode(errors="replace")))
        self.desc = d

    def close(self):
        if getattr(self, "_h", None):
            lib().kry_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def comm_init(self, uid, rank, nranks):
        buf = C.create_string_buffer(bytes(uid), KRY_UID_BYTES)
        _check(lib().kry_comm_init(self._h, buf, rank, nranks), "kry_comm_init")

    def comm_init_callbacks(self, rank, nranks, allreduce, halo):
        """allreduce(dev_ptr, n) and halo(peer_lo, send_lo, recv_lo, peer_hi, send_hi, recv_hi, count)
        are Python callables taking device addresses (ints); they return nothing."""
        def ar(ctx, buf, n):
            try:
                allreduce(buf, n)
                return 0
            except Exception:          # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1

        def hl(ctx, plo, slo, rlo, phi_, shi, rhi, count):
            try:
                halo(plo, slo, rlo, phi_, shi, rhi, count)
                return 0
            except Exception:          # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1
        self._cbs = (ALLREDUCE_CB(ar), HALO_CB(hl))      # keep alive
        _check(lib().kry_comm_init_callbacks(self._h, rank, nranks, self._cbs[0], self._cbs[1], None),
               "kry_comm_init_callbacks")

    def set_profile(self, on):
        _check(lib().kry_set_profile(self._h, int(bool(on))), "kry_set_profile")

    def mail_export(self):
        """This rank's mailbox handle (bytes) for the device-side all-reduce."""
        buf = C.create_string_buffer(KRY_MAIL_HANDLE_BYTES)
        _check(lib().kry_mail_export(self._h, buf), "kry_mail_export")
        return buf.raw

    def mail_connect(self, rank, nranks, handles):
        """`handles`: the ranks' mail_export() results in rank order."""
        blob = b"".join(bytes(x) for x in handles)
        assert len(blob) == nranks * KRY_MAIL_HANDLE_BYTES
        buf = C.create_string_buffer(blob, len(blob))
        _check(lib().kry_mail_connect(self._h, rank, nranks, buf), "kry_mail_connect")

    def mail_enable(self, what):
        """0: everything through the transport; KRY_MAIL_SUMS: dot products through the mailboxes;
        KRY_MAIL_SUMS | KRY_MAIL_PLANES: the ghost planes through mapped memory as well."""
        _check(lib().kry_mail_enable(self._h, int(what)), "kry_mail_enable")

    def solve_const(self, phi, rhs, lam, Dc, debug=0):
        return _check(lib().kry_solve_const(self._h, phi, rhs, lam, Dc[0], Dc[1], Dc[2], debug),
                      "kry_solve_const")

    def solve_var(self, phi, rhs, lam, Dxx, Dyy, Dzz, Dxy=None, use_offdiag=False, debug=0):
        return _check(lib().kry_solve_var(self._h, phi, rhs, lam, Dxx, Dyy, Dzz, _ptr(Dxy),
                                          int(use_offdiag), debug), "kry_solve_var")

    def load_const(self, phi, rhs, lam, Dc):
        return _check(lib().kry_load_const(self._h, phi, rhs, lam, Dc[0], Dc[1], Dc[2]),
                      "kry_load_const")

    def load_var(self, phi, rhs, lam, Dxx, Dyy, Dzz, Dxy=None, use_offdiag=False):
        return _check(lib().kry_load_var(self._h, phi, rhs, lam, Dxx, Dyy, Dzz, _ptr(Dxy),
                                         int(use_offdiag)), "kry_load_var")

    def update_rhs(self, rhs):
        return _check(lib().kry_update_rhs(self._h, rhs), "kry_update_rhs")

    def update_phi(self, phi):
        return _check(lib().kry_update_phi(self._h, phi), "kry_update_phi")

    def update_lambda(self, lam):
        return _check(lib().kry_update_lambda(self._h, lam), "kry_update_lambda")

    def reset(self):
        return _check(lib().kry_reset(self._h), "kry_reset")

    def resolve(self, phi=None, debug=0):
        """Prologue + iteration to convergence on the resident fields; phi (optional) receives the solution."""
        return _check(lib().kry_resolve(self._h, _ptr(phi), debug), "kry_resolve")

    def iterate(self, n):
        return _check(lib().kry_iterate(self._h, int(n)), "kry_iterate")

    def fetch_phi(self, phi):
        return _check(lib().kry_fetch_phi(self._h, phi), "kry_fetch_phi")

    def fetch_residual(self, r0):
        return _check(lib().kry_fetch_residual(self._h, r0), "kry_fetch_residual")

    def slab_link_note(self):
        return lib().kry_slab_link_note(self._h).decode(errors="replace")

    def stats(self):
        s = KryStats()
        _check(lib().kry_get_stats(self._h, C.byref(s)), "kry_get_stats")
        return s

    def history(self, cap=65536):
        buf = np.zeros(cap, dtype=np.float64)
        n = _check(lib().kry_get_history(self._h, _ptr(buf), cap), "kry_get_history")
        return buf[:min(n, cap)].copy()

    def step_times(self, cap=65536):
        """Wall-clock milliseconds of every outer iteration since the last load."""
        buf = np.zeros(cap, dtype=np.float64)
        n = _check(lib().kry_get_step_times(self._h, _ptr(buf), cap), "kry_get_step_times")
        return buf[:min(n, cap)].copy()

    # single operators ------------------------------------------------------
    def op_do_bcs(self, field):
        _check(lib().kry_op_do_bcs(self._h, field), "kry_op_do_bcs")
        return field

    def op_spmv_const(self, psi, lam, Dc, apply_bcs=True, fused=False):
        out = np.zeros_like(psi)
        _check(lib().kry_op_spmv_const(self._h, psi, lam, Dc[0], Dc[1], Dc[2], out,
                                       int(apply_bcs), int(fused)), "kry_op_spmv_const")
        return out

    def op_spmv_var(self, psi, lam, Dxx, Dyy, Dzz, Dxy=None, use_offdiag=False, apply_bcs=True,
                    fused=False):
        out = np.zeros_like(psi)
        _check(lib().kry_op_spmv_var(self._h, psi, lam, Dxx, Dyy, Dzz, _ptr(Dxy), int(use_offdiag),
                                     out, int(apply_bcs), int(fused)), "kry_op_spmv_var")
        return out

    def op_dot(self, a, b):
        r = C.c_double(0)
        _check(lib().kry_op_dot(self._h, a, b, C.byref(r)), "kry_op_dot")
        return r.value

    def op_l2norm(self, v):
        r = C.c_double(0)
        _check(lib().kry_op_l2norm(self._h, v, C.byref(r)), "kry_op_l2norm")
        return r.value


def solve_problem(p, L=2, eps=1e-6, max_outer=0, flags=0, nslabs_local=1, devices=(0,), debug=0):
    """Solve a krylov_amd.problems.Problem on this process' device(s)."""
    with Solver(p.mx, p.delx, p.bc, L=L, eps=eps, max_outer=max_outer, flags=flags,
                nslabs=nslabs_local, slab_begin=0, slab_count=nslabs_local, devices=devices) as s:
        phi = p.phi.copy()
        if p.var_coef:
            st = s.solve_var(phi, p.rhs, p.lam, p.Dxx, p.Dyy, p.Dzz, p.Dxy, p.use_offdiag, debug)
        else:
            st = s.solve_const(phi, p.rhs, p.lam, p.Dc, debug)
        stats = s.stats()
        hist = s.history()
        note = s.slab_link_note()
    return dict(status=st, phi=phi, history=hist, iters=stats.iterations, norm_rhs=stats.norm_rhs,
                slab_link=int(stats.slab_link), slab_link_note=note,
                norm_init=stats.norm_init, norm_final=stats.norm_final, fused=bool(stats.fused), path=int(stats.fused),
                solve_ms=stats.solve_ms, stage_ms=stats.stage_ms,
                order=(stats.tile_rows, stats.chunk_planes, stats.stream_blocks))
