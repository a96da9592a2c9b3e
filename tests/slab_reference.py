"""A z-slab-decomposed BiCGStab(l) written with the CPU oracle's operators and
an abstract communicator.  It follows the communication schedule of the HIP
library (krylov_amd/csrc/kry_solver.cpp):

  * every slab keeps one ghost plane on each interface; whole-array vector
    updates run on the ghosts as well, so a vector that is a linear combination
    of vectors with valid ghosts has valid ghosts without any exchange;
  * only operator OUTPUTS are exchanged (plane 1 to the lower neighbour's ghost
    mz+1, plane mz-1 to the upper neighbour's ghost -1), once per application;
  * every dot product is a local weighted sum followed by one all-reduce.

Interface planes are shared by both neighbours and carry weight 1/2 on each
side, as in the reference's MPI build (BICGStab3D/Krylov.C:180-201).
Test infrastructure only.
"""
import numpy as np


class SlabSolver:
    def __init__(self, orc, comm, rank, nranks, p_local, L, eps, max_outer):
        self.orc, self.comm, self.rank, self.n = orc, comm, rank, nranks
        self.p, self.L, self.eps, self.max_outer = p_local, L, eps, max_outer
        bc = list(p_local.bc)
        if rank > 0:
            bc[4] = -1          # interface: do_BCs leaves it alone
        if rank < nranks - 1:
            bc[5] = -1
        self.cfg = orc.make_cfg(p_local.mx, p_local.delx, bc, var_coef=p_local.var_coef,
                                use_offdiag=p_local.use_offdiag, Dc=p_local.Dc)

    # -- communication ----------------------------------------------------
    def halo(self, a):
        mz = self.p.mx[2]
        lo = self.rank - 1 if self.rank > 0 else None
        hi = self.rank + 1 if self.rank < self.n - 1 else None
        got_lo, got_hi = self.comm.exchange(lo, a[2].copy(), hi, a[mz].copy())
        if got_lo is not None:
            a[0] = got_lo       # ghost plane -1
        if got_hi is not None:
            a[mz + 2] = got_hi  # ghost plane mz+1

    def dot(self, a, b):
        return self.comm.allreduce(self.orc.dot(self.cfg, a, b))

    def norm(self, a):
        # unscaled weighted sum of squares; the eight corners of the GLOBAL domain
        # weigh 1/4 in the reference's norm instead of the dot's 1/8 (Krylov.C:136-143)
        s = self.orc.dot(self.cfg, a, a)
        mx, my, mz = self.p.mx
        ks = ([1] if self.rank == 0 else []) + ([mz + 1] if self.rank == self.n - 1 else [])
        for k in ks:
            for j in (1, my + 1):
                for i in (1, mx + 1):
                    s += 0.125 * a[k, j, i] * a[k, j, i]
        return np.sqrt(self.comm.allreduce(s))

    # -- operator -----------------------------------------------------------
    def apply(self, psi, apply_bcs=True):
        p = self.p
        if p.var_coef:
            out = self.orc.spmv_var(self.cfg, psi, p.lam, p.Dxx, p.Dyy, p.Dzz, p.Dxy, apply_bcs)
        else:
            out = self.orc.spmv_const(self.cfg, psi, p.lam, apply_bcs)
        self.halo(out)
        return out

    def solve(self):
        p, L, eps = self.p, self.L, self.eps
        phi = p.phi.copy()
        for f in (p.Dxx, p.Dyy, p.Dzz, p.Dxy, phi):
            if f is not None:
                self.halo(f)
        norm_rhs = self.norm(p.rhs) or 1.0
        r = [np.zeros_like(phi) for _ in range(L + 1)]
        u = [np.zeros_like(phi) for _ in range(L + 1)]
        r[0] = p.rhs - self.apply(phi, apply_bcs=False)
        self.orc.do_bcs(self.cfg, r[0])
        self.halo(r[0])
        hist = []
        if self.norm(r[0]) < eps * norm_rhs:
            return phi, hist, norm_rhs
        rt = r[0].copy()
        rho0, alpha, omega = 1.0, 0.0, 1.0
        tau = np.zeros((L + 1, L + 1))
        sigma, gp, g, gpp = (np.zeros(L + 1) for _ in range(4))
        while True:
            rho0 *= -omega
            for jj in range(L):
                rho1 = self.dot(r[jj], rt)
                beta = alpha * rho1 / rho0
                rho0 = rho1
                for ii in range(jj + 1):
                    u[ii] = u[ii] * (-beta) + r[ii]
                u[jj + 1] = self.apply(u[jj])
                alpha = rho0 / self.dot(u[jj + 1], rt)
                for ii in range(jj + 1):
                    r[ii] = r[ii] - u[ii + 1] * alpha
                r[jj + 1] = self.apply(r[jj])
                phi = phi + u[0] * alpha
            for jj in range(1, L + 1):
                for ii in range(1, jj):
                    tau[ii][jj] = self.dot(r[jj], r[ii]) / sigma[ii]
                    r[jj] = r[jj] - r[ii] * tau[ii][jj]
                sigma[jj] = self.dot(r[jj], r[jj])
                gp[jj] = self.dot(r[0], r[jj]) / sigma[jj]
            omega = g[L] = gp[L]
            for jj in range(L - 1, 0, -1):
                g[jj] = gp[jj]
                for ii in range(jj + 1, L + 1):
                    g[jj] -= tau[jj][ii] * g[ii]
            for jj in range(1, L):
                gpp[jj] = g[jj + 1]
                for ii in range(jj + 1, L):
                    gpp[jj] += tau[jj][ii] * g[ii + 1]
            phi = phi + r[0] * g[1]
            r[0] = r[0] - r[L] * gp[L]
            u[0] = u[0] - u[L] * g[L]
            for jj in range(1, L):
                u[0] = u[0] - u[jj] * g[jj]
                phi = phi + r[jj] * gpp[jj]
                r[0] = r[0] - r[jj] * gp[jj]
            nrm = self.norm(r[0])
            hist.append(nrm)
            if not (nrm > eps * norm_rhs) or (self.max_outer and len(hist) >= self.max_outer):
                break
        return phi, hist, norm_rhs


def split_problem(P, p, nranks, rank):
    """Slab `rank` of a global Problem, ghost planes taken from the global fields."""
    mzl = p.mx[2] // nranks
    sl = slice(rank * mzl, rank * mzl + mzl + 3)
    cut = lambda a: None if a is None else np.ascontiguousarray(a[sl])
    return P.Problem(p.name, (p.mx[0], p.mx[1], mzl), p.delx, p.bc, p.var_coef, p.use_offdiag, p.Dc,
                     cut(p.phi), cut(p.rhs), cut(p.lam), cut(p.Dxx), cut(p.Dyy), cut(p.Dzz), cut(p.Dxy))


def split_brick(P, p, nproc, coord):
    """Sub-domain `coord` of an nproc[0] x nproc[1] x nproc[2] Cartesian split of a
    global Problem, ghost layers taken from the global fields (arrays are z, y, x)."""
    m = [p.mx[d] // nproc[d] for d in range(3)]
    sx = slice(coord[0] * m[0], coord[0] * m[0] + m[0] + 3)
    sy = slice(coord[1] * m[1], coord[1] * m[1] + m[1] + 3)
    sz = slice(coord[2] * m[2], coord[2] * m[2] + m[2] + 3)
    cut = lambda a: None if a is None else np.ascontiguousarray(a[sz, sy, sx])
    return P.Problem(p.name, tuple(m), p.delx, p.bc, p.var_coef, p.use_offdiag, p.Dc,
                     cut(p.phi), cut(p.rhs), cut(p.lam), cut(p.Dxx), cut(p.Dyy), cut(p.Dzz), cut(p.Dxy))


class BrickSolver(SlabSolver):
    """The same schedule on a Cartesian split (sub-domains in x, y and z, the layout of
    the reference's mpi_manager_3D): operator outputs are exchanged x first, then y,
    then z, each over the full tangential box, so edges and corners travel in two or
    three hops (BICGStab3D/BoundaryHandler.C:579-1045).  rank = (cx*ny + cy)*nz + cz."""

    def __init__(self, orc, comm, nproc, coord, p_local, L, eps, max_outer):
        self.orc, self.comm, self.np3, self.co = orc, comm, tuple(nproc), tuple(coord)
        self.p, self.L, self.eps, self.max_outer = p_local, L, eps, max_outer
        bc = list(p_local.bc)
        for d in range(3):
            if coord[d] > 0:
                bc[2 * d] = -1
            if coord[d] < nproc[d] - 1:
                bc[2 * d + 1] = -1
        self.cfg = orc.make_cfg(p_local.mx, p_local.delx, bc, var_coef=p_local.var_coef,
                                use_offdiag=p_local.use_offdiag, Dc=p_local.Dc)

    def _rank(self, c):
        return (c[0] * self.np3[1] + c[1]) * self.np3[2] + c[2]

    def halo(self, a):
        for d in range(3):                      # arrays are indexed [z, y, x]
            ax = 2 - d
            m = self.p.mx[d]
            c = list(self.co)
            lo = hi = None
            if self.co[d] > 0:
                c[d] -= 1; lo = self._rank(c); c[d] += 1
            if self.co[d] < self.np3[d] - 1:
                c[d] += 1; hi = self._rank(c); c[d] -= 1
            if lo is None and hi is None:
                continue
            take = lambda i: np.ascontiguousarray(np.take(a, i, axis=ax))
            got_lo, got_hi = self.comm.exchange(lo, take(2), hi, take(m))
            idx = [slice(None)] * 3
            if got_lo is not None:
                idx[ax] = 0; a[tuple(idx)] = got_lo
            if got_hi is not None:
                idx[ax] = m + 2; a[tuple(idx)] = got_hi

    def norm(self, a):
        s = self.orc.dot(self.cfg, a, a)
        mx, my, mz = self.p.mx
        on = lambda d, side: self.co[d] == (0 if side == 0 else self.np3[d] - 1)
        for kz, k in ((0, 1), (1, mz + 1)):
            for jy, j in ((0, 1), (1, my + 1)):
                for ix, i in ((0, 1), (1, mx + 1)):
                    if on(2, kz) and on(1, jy) and on(0, ix):       # a corner of the GLOBAL domain
                        s += 0.125 * a[k, j, i] * a[k, j, i]
        return np.sqrt(self.comm.allreduce(s))
