"""TEST INFRASTRUCTURE ONLY -- a second, independently written restatement of the serial
reference path, in numpy, used to cross-check oracle/krylov_oracle.c.

Written from the reference's source text alone (not from the C restatement):
  BICGStab3D/solveLin_BICGStab.C:200-215, 218-236, 250-527   solve, solve_int
  BICGStab3D/Krylov.C:36-157, 161-244, 248-306, 311-402,
                      407-496, 500-541, 553-559              norm, dot, residual, operators, reallyIsNan
  BICGStab3D/BoundaryHandler.C:370-575                       do_BCs, serial build, rim = 1
Arrays are numpy float64 of shape (mz+3, my+3, mx+3): element [k+1, j+1, i+1] is the
reference's dist(i, j, k), i in [-1, mx+1].

Every elementwise numpy operation rounds once, as the reference's scalar expression does
(no fused multiply-add), and every sum runs sequentially in the reference's loop order
(numpy's add.accumulate is a plain left-to-right loop; np.sum is pairwise and is NOT
used).  The two restatements must therefore agree bit for bit; that is what
tests/test_oracle_second_restatement.py checks.  Parity with the reference itself stays
unpinned (the reference holds no numeric fixtures and cannot be built here)."""
import numpy as np


def _seq(acc, terms):
    """acc + t0 + t1 + ... added one by one, left to right."""
    t = np.ascontiguousarray(terms, dtype=np.float64).ravel()
    if t.size == 0:
        return acc
    return float(np.add.accumulate(np.concatenate(([acc], t)))[-1])


def _sqr(x):
    return x * x


class Grid:
    def __init__(self, mx, delx, bc, Dc=(1., 1., 1.)):
        self.mx = tuple(int(m) for m in mx)
        self.delx = tuple(float(d) for d in delx)
        self.bc = tuple(int(b) for b in bc)
        self.Dc = tuple(float(d) for d in Dc)


def at(a, i, j, k):
    """dist(i, j, k) with slices or ints in the reference's index space (shifted by the rim)."""
    sh = lambda s: slice(s.start + 1, s.stop + 1) if isinstance(s, slice) else s + 1
    return a[sh(k), sh(j), sh(i)]


# ---------------------------------------------------------------- BoundaryHandler.C:370-575
def do_bcs(g, d):
    mx, my, mz = g.mx
    Y, Z = slice(-1, my + 2), slice(-1, mz + 2)                           # [-rim, m+rim]
    PX, PY = slice(0, mx + 1), slice(0, my + 1)                           # [0, m]
    # lower x (:417-440)
    if g.bc[0] == 0:
        at(d, slice(-1, 1), Y, Z)[...] = 0.
    elif g.bc[0] == 1:
        at(d, -1, Y, Z)[...] = 2. * at(d, 0, Y, Z) - at(d, 1, Y, Z)
    # upper x (:443-462)
    if g.bc[1] == 0:
        at(d, slice(mx, mx + 2), Y, Z)[...] = 0.
    elif g.bc[1] == 1:
        for ipos in (mx, mx + 1):
            at(d, ipos, Y, Z)[...] = 2. * at(d, ipos - 1, Y, Z) - at(d, ipos - 2, Y, Z)
    # lower y (:477-496)
    if g.bc[2] == 0:
        at(d, PX, slice(-1, 1), Z)[...] = 0.
    elif g.bc[2] == 1:
        at(d, PX, -1, Z)[...] = 2. * at(d, PX, 0, Z) - at(d, PX, 1, Z)
    # upper y (:500-519)
    if g.bc[3] == 0:
        at(d, PX, slice(my, my + 2), Z)[...] = 0.
    elif g.bc[3] == 1:
        for ipos in (my, my + 1):
            at(d, PX, ipos, Z)[...] = 2. * at(d, PX, ipos - 1, Z) - at(d, PX, ipos - 2, Z)
    # lower z (:527-546): ipos runs -1, 0 in this order also for the extrapolation
    if g.bc[4] == 0:
        at(d, PX, PY, slice(-1, 1))[...] = 0.
    elif g.bc[4] == 1:
        for ipos in (-1, 0):
            at(d, PX, PY, ipos)[...] = 2. * at(d, PX, PY, ipos + 1) - at(d, PX, PY, ipos + 2)
    # upper z (:549-568)
    if g.bc[5] == 0:
        at(d, PX, PY, slice(mz, mz + 2))[...] = 0.
    elif g.bc[5] == 1:
        for ipos in (mz, mz + 1):
            at(d, PX, PY, ipos)[...] = 2. * at(d, PX, PY, ipos - 1) - at(d, PX, PY, ipos - 2)
    return d


# ---------------------------------------------------------------- Krylov.C:36-157
def l2norm(g, v):
    mx, my, mz = g.mx
    IX, IY, IZ = slice(1, mx), slice(1, my), slice(1, mz)
    val_max = float(np.max(np.abs(at(v, slice(0, mx + 1), slice(0, my + 1), slice(0, mz + 1)))))   # :49-57
    if val_max == 0.:
        return 0.
    s = 1. / val_max
    q = lambda i, j, k: _sqr(s * at(v, i, j, k))
    total = _seq(0., q(IX, IY, IZ))                                                        # :81-88
    # faces, two terms per inner step (:92-111); arrays come out as [z, y(, x)]
    total = _seq(total, np.stack((0.5 * q(0, IY, IZ), 0.5 * q(mx, IY, IZ)), axis=-1))
    total = _seq(total, np.stack((0.5 * q(IX, 0, IZ), 0.5 * q(IX, my, IZ)), axis=-1))
    total = _seq(total, np.stack((0.5 * q(IX, IY, 0), 0.5 * q(IX, IY, mz)), axis=-1))
    # edges (:115-133)
    total = _seq(total, np.stack((0.25 * q(0, 0, IZ), 0.25 * q(0, my, IZ), 0.25 * q(mx, 0, IZ), 0.25 * q(mx, my, IZ)), axis=-1))
    total = _seq(total, np.stack((0.25 * q(0, IY, 0), 0.25 * q(0, IY, mz), 0.25 * q(mx, IY, 0), 0.25 * q(mx, IY, mz)), axis=-1))
    total = _seq(total, np.stack((0.25 * q(IX, 0, 0), 0.25 * q(IX, 0, mz), 0.25 * q(IX, my, 0), 0.25 * q(IX, my, mz)), axis=-1))
    # corners, weight 0.25 as written (:136-143)
    for (i, j, k) in ((0, 0, 0), (0, my, 0), (0, 0, mz), (0, my, mz), (mx, 0, 0), (mx, my, 0), (mx, 0, mz), (mx, my, mz)):
        total = total + 0.25 * float(q(i, j, k))
    return val_max * float(np.sqrt(total))


# ---------------------------------------------------------------- Krylov.C:161-244
def dot(g, a, b):
    mx, my, mz = g.mx
    IX, IY, IZ = slice(1, mx), slice(1, my), slice(1, mz)
    p = lambda w, i, j, k: (w * at(a, i, j, k)) * at(b, i, j, k)       # 0.5*vecA(..)*vecB(..) groups to the left
    total = _seq(0., at(a, IX, IY, IZ) * at(b, IX, IY, IZ))            # :171-177
    total = _seq(total, np.stack((p(0.5, 0, IY, IZ), p(0.5, mx, IY, IZ)), axis=-1))        # :182-187
    total = _seq(total, np.stack((p(0.5, IX, 0, IZ), p(0.5, IX, my, IZ)), axis=-1))        # :189-194
    total = _seq(total, np.stack((p(0.5, IX, IY, 0), p(0.5, IX, IY, mz)), axis=-1))        # :196-201
    total = _seq(total, np.stack((p(0.25, 0, 0, IZ), p(0.25, mx, 0, IZ), p(0.25, 0, my, IZ), p(0.25, mx, my, IZ)), axis=-1))
    total = _seq(total, np.stack((p(0.25, 0, IY, 0), p(0.25, mx, IY, 0), p(0.25, 0, IY, mz), p(0.25, mx, IY, mz)), axis=-1))
    total = _seq(total, np.stack((p(0.25, IX, 0, 0), p(0.25, IX, my, 0), p(0.25, IX, 0, mz), p(0.25, IX, my, mz)), axis=-1))
    for (i, j, k) in ((0, 0, 0), (mx, 0, 0), (0, my, 0), (mx, my, 0), (0, 0, mz), (mx, 0, mz), (0, my, mz), (mx, my, mz)):
        total = total + float(p(0.125, i, j, k))                       # :226-233
    return total


# ---------------------------------------------------------------- Krylov.C:407-496
def multiply_var(g, psi, lam, Dxx, Dyy, Dzz, Dxy, out, apply_bcs):
    mx, my, mz = g.mx
    dx, dy, dz = g.delx
    c0, c1, c2 = 1. / _sqr(dx), 1. / _sqr(dy), 1. / _sqr(dz)
    cxy = 1. / (2. * dx * dy)

    def s(a, di=0, dj=0, dk=0):
        return at(a, slice(di, mx + 1 + di), slice(dj, my + 1 + dj), slice(dk, mz + 1 + dk))
    P = lambda di=0, dj=0, dk=0: s(psi, di, dj, dk)
    v = c0 * s(Dxx) * (P(1) + P(-1))
    v = v + c1 * s(Dyy) * (P(0, 1) + P(0, -1))
    v = v + c2 * s(Dzz) * (P(0, 0, 1) + P(0, 0, -1))
    v = v - (2. * (c0 * s(Dxx) + c1 * s(Dyy) + c2 * s(Dzz)) + s(lam)) * P()
    v = v + cxy * s(Dxy) * (P(1, 1) - P(1, -1) - P(-1, 1) + P(-1, -1))
    v = v + ((s(Dxx, 1) - s(Dxx, -1)) / (2. * dx) + (s(Dxy, 0, 1) - s(Dxy, 0, -1)) / (2. * dy)) * (P(1) - P(-1)) / (2. * dx)
    v = v + ((s(Dxy, 1) - s(Dxy, -1)) / (2. * dx) + (s(Dyy, 0, 1) - s(Dyy, 0, -1)) / (2. * dy)) * (P(0, 1) - P(0, -1)) / (2. * dy)
    v = v + ((s(Dzz, 0, 0, 1) - s(Dzz, 0, 0, -1)) / (2. * dz)) * (P(0, 0, 1) - P(0, 0, -1)) / (2. * dz)
    s(out)[...] = v
    if apply_bcs:
        do_bcs(g, out)


# ---------------------------------------------------------------- Krylov.C:500-541
def multiply_const(g, psi, lam, out, apply_bcs):
    mx, my, mz = g.mx
    c0, c1, c2 = (g.Dc[d] / _sqr(g.delx[d]) for d in range(3))
    do_bcs(g, psi)                                                     # :516, on the input

    def P(di=0, dj=0, dk=0):
        return at(psi, slice(di, mx + 1 + di), slice(dj, my + 1 + dj), slice(dk, mz + 1 + dk))
    lamp = at(lam, slice(0, mx + 1), slice(0, my + 1), slice(0, mz + 1))
    v = c0 * (P(1) + P(-1)) + c1 * (P(0, 1) + P(0, -1)) + c2 * (P(0, 0, 1) + P(0, 0, -1)) \
        - (2. * (c0 + c1 + c2) + lamp) * P()
    at(out, slice(0, mx + 1), slice(0, my + 1), slice(0, mz + 1))[...] = v
    if apply_bcs:
        do_bcs(g, out)


def really_is_nan(x):
    """Krylov.C:553-559: the argument is narrowed to float first."""
    with np.errstate(over="ignore", invalid="ignore"):
        bits = np.array([x], dtype=np.float64).astype(np.float32).view(np.int32)[0]
    return (int(bits) & 0x7F800000) == 0x7F800000


# ---------------------------------------------------------------- solveLin_BICGStab.C:250-527
def solve(g, L, eps, phi, rhs, lam, Dxx=None, Dyy=None, Dzz=None, Dxy=None, use_offdiag=False, max_outer=0):
    """phi is updated in place.  Returns a dict with the history of the norm after every
    pass.  max_outer > 0 stops after that many passes (the reference has no cap)."""
    var = Dxx is not None
    if var and not use_offdiag:
        Dxy = np.zeros_like(rhs)                                       # :231-233, Dxy.clear()

    def A(psi, out, apply_bcs):
        if var:
            multiply_var(g, psi, lam, Dxx, Dyy, Dzz, Dxy, out, apply_bcs)
        else:
            multiply_const(g, psi, lam, out, apply_bcs)

    def residual(out):                                                 # Krylov.C:248-306, 311-402
        A(phi, out, False)
        out[...] = rhs - out
        do_bcs(g, out)

    r = [np.zeros_like(rhs) for _ in range(L + 1)]                     # :180-189, zeroed once
    u = [np.zeros_like(rhs) for _ in range(L + 1)]
    res = dict(history=[], iters=0, status=0)
    norm_rhs = l2norm(g, rhs)
    if norm_rhs == 0.:
        norm_rhs = 1.
    residual(r[0])
    norm_init = l2norm(g, r[0])
    res.update(norm_rhs=norm_rhs, norm_init=norm_init)
    if norm_init < eps * norm_rhs:
        res["norm_final"] = norm_init
        return res
    rt = r[0].copy()
    norm, rho0, alpha, omega = 1.e99, 1., 0., 1.
    tau = np.zeros((L + 1, L + 1))
    sigma, gammap, gammapp, gamma = (np.zeros(L + 1) for _ in range(4))
    it = 0
    while True:
        it += 1
        rho0 *= -omega
        for jj in range(L):                                            # :345-410
            rho1 = dot(g, r[jj], rt)
            beta = alpha * rho1 / rho0
            rho0 = rho1
            for ii in range(jj + 1):
                u[ii] *= -beta
                u[ii] += r[ii]
            A(u[jj], u[jj + 1], True)
            alpha = rho0 / dot(g, u[jj + 1], rt)
            for ii in range(jj + 1):
                r[ii] -= u[ii + 1] * alpha
            A(r[jj], r[jj + 1], True)
            phi += u[0] * alpha
            if really_is_nan(alpha):
                res.update(status=1, iters=it, norm_final=norm)
                return res
        for jj in range(1, L + 1):                                     # :419-430
            for ii in range(1, jj):
                tau[ii][jj] = dot(g, r[jj], r[ii]) / sigma[ii]
                r[jj] -= r[ii] * tau[ii][jj]
            sigma[jj] = dot(g, r[jj], r[jj])
            gammap[jj] = dot(g, r[0], r[jj]) / sigma[jj]
        omega = gamma[L] = gammap[L]
        for jj in range(L - 1, 0, -1):                                 # :434-439
            gamma[jj] = gammap[jj]
            for ii in range(jj + 1, L + 1):
                gamma[jj] -= tau[jj][ii] * gamma[ii]
        for jj in range(1, L):                                         # :441-446
            gammapp[jj] = gamma[jj + 1]
            for ii in range(jj + 1, L):
                gammapp[jj] += tau[jj][ii] * gamma[ii + 1]
        phi += r[0] * gamma[1]                                         # :450-462
        r[0] -= r[L] * gammap[L]
        u[0] -= u[L] * gamma[L]
        for jj in range(1, L):
            u[0] -= u[jj] * gamma[jj]
            phi += r[jj] * gammapp[jj]
            r[0] -= r[jj] * gammap[jj]
        norm = l2norm(g, r[0])
        res["history"].append(norm)
        if really_is_nan(norm):
            res.update(status=3, iters=it, norm_final=norm)
            return res
        if not (norm > eps * norm_rhs):
            break
        if max_outer > 0 and it >= max_outer:
            res["status"] = 4
            break
    res.update(iters=it, norm_final=norm)
    residual(r[0])                                                     # :512-518, result unused; the constant
    #                                                                    operator's input treatment touches phi
    return res
