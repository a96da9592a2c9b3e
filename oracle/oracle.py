"""ctypes binding of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Loaded by tests/, by __graft_entry__.smoke() and by the cpu_baseline leg of
bench.py.  Nothing under krylov_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libkrylov_oracle.so")


class OrcCfg(C.Structure):
    _fields_ = [
        ("mx", C.c_int * 3),
        ("delx", C.c_double * 3),
        ("bc", C.c_int * 6),
        ("L", C.c_int),
        ("eps", C.c_double),
        ("max_outer", C.c_int),
        ("var_coef", C.c_int),
        ("use_offdiag", C.c_int),
        ("Dc", C.c_double * 3),
        ("debug", C.c_int),
    ]


class OrcOut(C.Structure):
    _fields_ = [
        ("iters", C.c_int),
        ("status", C.c_int),
        ("norm_rhs", C.c_double),
        ("norm_init", C.c_double),
        ("norm_final", C.c_double),
        ("norm_true", C.c_double),
    ]


def build(force=False):
    """Compile the oracle with the committed Makefile (gcc only)."""
    srcs = [os.path.join(_HERE, f) for f in ("krylov_oracle.c", "gpu_order.c", "krylov_oracle_ranks.c", "krylov_oracle.h")]
    if (not force and os.path.exists(_LIB)
            and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in srcs)):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None
_OMP_LIB = os.path.join(_HERE, "libkrylov_oracle_omp.so")


def use_openmp_build():
    """Switch this module to the -fopenmp build of the same sources (CPU
    baseline timing only; its sums are not in the reference's order)."""
    global _lib, _LIB
    if not os.path.exists(_OMP_LIB):
        build(force=True)
    _LIB, _lib = _OMP_LIB, None


_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        P = C.POINTER
        L.orc_solve.restype = C.c_int
        L.orc_solve.argtypes = [P(OrcCfg), _dp, _dp, _dp, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, _dp, C.c_int, P(OrcOut)]
        L.orc_solve_ranks.restype = C.c_int
        L.orc_solve_ranks.argtypes = [P(OrcCfg), C.c_int * 3, _dp, _dp, _dp, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, _dp, C.c_int, P(OrcOut)]
        L.orc_do_bcs.restype = None
        L.orc_do_bcs.argtypes = [P(OrcCfg), _dp]
        L.orc_spmv_const.restype = None
        L.orc_spmv_const.argtypes = [P(OrcCfg), _dp, _dp, _dp, C.c_int]
        L.orc_spmv_var.restype = None
        L.orc_spmv_var.argtypes = [P(OrcCfg), _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int]
        L.orc_dot.restype = C.c_double
        L.orc_dot.argtypes = [P(OrcCfg), _dp, _dp]
        L.orc_l2norm.restype = C.c_double
        L.orc_l2norm.argtypes = [P(OrcCfg), _dp]
        L.orc_really_is_nan.restype = C.c_int
        L.orc_really_is_nan.argtypes = [C.c_double]
        L.orc_use_gpu_order.restype = None
        L.orc_use_gpu_order.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_gpu_order_slabs.restype = None
        L.orc_gpu_order_slabs.argtypes = [C.c_int, C.c_int]
        L.orc_store_size.restype = C.c_long
        L.orc_store_size.argtypes = [P(OrcCfg)]
        L.orc_hash.restype = C.c_ulonglong
        L.orc_hash.argtypes = [P(OrcCfg), _dp]
        L.orc_set_hash_trace.restype = None
        L.orc_set_hash_trace.argtypes = [C.c_void_p, C.c_int]
        L.orc_hash_trace_count.restype = C.c_int
        L.orc_hash_trace_count.argtypes = []
        _lib = L
    return _lib


def make_cfg(mx, delx, bc=(0,) * 6, L=2, eps=1e-6, max_outer=0, var_coef=False,
             use_offdiag=False, Dc=(1., 1., 1.), debug=0):
    c = OrcCfg()
    for d in range(3):
        c.mx[d] = int(mx[d])
        c.delx[d] = float(delx[d])
        c.Dc[d] = float(Dc[d])
    for f in range(6):
        c.bc[f] = int(bc[f])
    c.L = int(L)
    c.eps = float(eps)
    c.max_outer = int(max_outer)
    c.var_coef = int(bool(var_coef))
    c.use_offdiag = int(bool(use_offdiag))
    c.debug = int(debug)
    return c


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def hash_field(cfg, a):
    """The reference's hash() of a field (BICGStab3D/solveLin_BICGStab.C:25-47): XOR of the raw bits of
    every stored cell, started from 797003437."""
    return int(lib().orc_hash(C.byref(cfg), a))


def solve(cfg, phi, rhs, lam, Dxx=None, Dyy=None, Dzz=None, Dxy=None, hist_cap=4096, hashes=False):
    """Run the oracle solve.  phi is updated in place.  Arrays are the
    ghost-inclusive boxes, shape (mz+3, my+3, mx+3), C order (x fastest).
    hashes: also record the reference's hash() at its own (disabled) print sites:
    result["input_hashes"] = (phi, rhs, lambda) on entry (:284-287), result["pass_hashes"] =
    [(phi, residuals[0]), ...] at the end of every pass (:473-477)."""
    hist = np.zeros(max(hist_cap, 1), dtype=np.float64)
    out = OrcOut()
    tr = np.zeros(3 + 2 * max(hist_cap, 1), dtype=np.uint64) if hashes else None
    if hashes:
        lib().orc_set_hash_trace(tr.ctypes.data_as(C.c_void_p), len(tr))
    try:
        st = lib().orc_solve(C.byref(cfg), phi, rhs, lam, _ptr(Dxx), _ptr(Dyy), _ptr(Dzz),
                             _ptr(Dxy), hist, hist_cap, C.byref(out))
        ntr = lib().orc_hash_trace_count() if hashes else 0
    finally:
        if hashes:
            lib().orc_set_hash_trace(None, 0)
    n = min(out.iters, hist_cap)
    res = dict(status=st, iters=out.iters, norm_rhs=out.norm_rhs, norm_init=out.norm_init,
               norm_final=out.norm_final, norm_true=out.norm_true, history=hist[:n].copy())
    if hashes:
        v = [int(x) for x in tr[:min(ntr, len(tr))]]
        res["input_hashes"] = tuple(v[:3])
        res["pass_hashes"] = [(v[i], v[i + 1]) for i in range(3, len(v) - 1, 2)]
    return res


def solve_ranks(cfg, nproc, phi, rhs, lam, Dxx=None, Dyy=None, Dzz=None, Dxy=None, hist_cap=4096):
    """The solve as the reference's MPI build runs it on nproc = (px, py, pz) sub-domains, one
    thread per rank (krylov_oracle_ranks.c).  Arrays are the GLOBAL boxes; phi is updated in place."""
    hist = np.zeros(max(hist_cap, 1), dtype=np.float64)
    out = OrcOut()
    st = lib().orc_solve_ranks(C.byref(cfg), (C.c_int * 3)(*nproc), phi, rhs, lam, _ptr(Dxx), _ptr(Dyy), _ptr(Dzz),
                               _ptr(Dxy), hist, hist_cap, C.byref(out))
    if st < 0:
        raise ValueError("the split %s does not divide the grid" % (tuple(nproc),))
    n = min(out.iters, hist_cap)
    return dict(status=st, iters=out.iters, norm_rhs=out.norm_rhs, norm_init=out.norm_init,
                norm_final=out.norm_final, norm_true=out.norm_true, history=hist[:n].copy())


def do_bcs(cfg, a):
    lib().orc_do_bcs(C.byref(cfg), a)


def spmv_const(cfg, psi, lam, apply_bcs=True):
    out = np.zeros_like(psi)
    lib().orc_spmv_const(C.byref(cfg), psi, lam, out, int(apply_bcs))
    return out


def spmv_var(cfg, psi, lam, Dxx, Dyy, Dzz, Dxy, apply_bcs=True):
    out = np.zeros_like(psi)
    lib().orc_spmv_var(C.byref(cfg), psi, lam, Dxx, Dyy, Dzz, Dxy, out, int(apply_bcs))
    return out


def dot(cfg, a, b):
    return lib().orc_dot(C.byref(cfg), a, b)


def l2norm(cfg, v):
    return lib().orc_l2norm(C.byref(cfg), v)


def use_gpu_order(mode, ty=8, zc=16, stream_blocks=2048):
    """mode 0: reference summation order; 1: order of the fused Dirichlet device
    path; 2: order of the generic device path; 3: order of the fused
    general-boundary device path (see gpu_order.c)."""
    lib().orc_use_gpu_order(int(mode), int(ty), int(zc), int(stream_blocks))


def gpu_order_slabs(nslabs=1, split_launch=False):
    """The device run to reproduce used `nslabs` z slabs; split_launch: with the
    edge/interior split of its stencil kernels (overlapped halo exchange)."""
    lib().orc_gpu_order_slabs(int(nslabs), int(bool(split_launch)))


def really_is_nan(x):
    return bool(lib().orc_really_is_nan(float(x)))
