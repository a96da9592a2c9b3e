"""Shared helpers of the parity tests: run a Problem through the CPU oracle."""
import numpy as np


def oracle_solve(orc, p, L=2, eps=1e-6, max_outer=0, hashes=False):
    cfg = orc.make_cfg(p.mx, p.delx, p.bc, L=L, eps=eps, max_outer=max_outer, var_coef=p.var_coef,
                       use_offdiag=p.use_offdiag, Dc=p.Dc)
    phi = p.phi.copy()
    cp = lambda a: None if a is None else a.copy()
    r = orc.solve(cfg, phi, p.rhs.copy(), p.lam.copy(), cp(p.Dxx), cp(p.Dyy), cp(p.Dzz), cp(p.Dxy), hashes=hashes)
    r["phi"] = phi
    return r


def rel_l2(a, b):
    d = (a - b).ravel()
    n = np.linalg.norm(b.ravel())
    return np.linalg.norm(d) / (n if n > 0 else 1.0)


def points(a):
    """Grid points 0..m (ghost layer stripped)."""
    return a[1:-1, 1:-1, 1:-1]
