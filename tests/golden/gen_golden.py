#!/usr/bin/env python3
"""Generates tests/golden/oracle_cases.json from the CPU oracle.

The reference holds no golden vectors and cannot be built here (its array
library matrix.H is not in its tree), so these fixtures pin the ORACLE, not the
reference: they freeze what oracle/krylov_oracle.c computes today so that a
later edit of the oracle, a different compiler or different host CPU shows up
as a diff.  Inputs use integer hashes and polynomials on dyadic grids only (no
libm), so they are bit-identical on every IEEE-754 machine.

Run from the repository root:  python tests/golden/gen_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402
from krylov_amd import problems as P  # noqa: E402


def hash_field(shape, salt):
    """Deterministic pseudo-random doubles in [-1, 1): exact integer arithmetic."""
    k, j, i = np.meshgrid(*[np.arange(n, dtype=np.uint64) for n in shape], indexing="ij")
    h = (i * np.uint64(73856093)) ^ (j * np.uint64(19349663)) ^ (k * np.uint64(83492791)) ^ np.uint64(salt * 2654435761)
    h = (h * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(43)          # 21 bits
    return h.astype(np.float64) / float(1 << 20) - 1.0


def make_case(mx, bc, var, offd, salt, Dc=(1.0, 2.0, 0.5)):
    mx = tuple(mx)
    delx = tuple(1.0 / (1 << int(np.ceil(np.log2(m)))) for m in mx)      # dyadic spacing
    x, y, z = P.coords(mx, delx)
    shp = x.shape
    p = P.Problem("golden", mx, delx, tuple(bc), var, offd, tuple(Dc),
                  P.box(mx), hash_field(shp, salt), 0.25 * x * y + 0.5 * z * z)
    if var:
        p.Dxx = 1.0 + 0.5 * x * y
        p.Dyy = 1.5 - 0.5 * y * z
        p.Dzz = 1.0 + 0.25 * z * x
        p.Dxy = 0.125 * x * (1.0 - y) * z if offd else P.box(mx)
    if all(b == 0 for b in bc):
        p.rhs = P.interior_only(p.rhs)
    return p


CASES = [
    # name, mx, bc, var, offd, L, max_outer
    ("const_dirichlet_l1", (16, 16, 16), (0,) * 6, False, False, 1, 50),
    ("const_dirichlet_l2", (16, 16, 16), (0,) * 6, False, False, 2, 50),
    ("const_dirichlet_l2_32", (32, 32, 32), (0,) * 6, False, False, 2, 50),
    ("var_dirichlet_l1", (16, 16, 16), (0,) * 6, True, False, 1, 50),
    ("var_dirichlet_l2", (16, 16, 16), (0,) * 6, True, False, 2, 50),
    ("var_offdiag_l2", (16, 16, 16), (0,) * 6, True, True, 2, 50),
    ("var_mixed_l2", (16, 16, 16), (0, 0, 1, 1, 0, 1), True, False, 2, 50),
    ("const_mixed_l1", (16, 16, 16), (1, 0, 0, 1, 1, 0), False, False, 1, 50),
    ("var_neumann_lowz_l4", (12, 10, 16), (0, 0, 0, 0, 1, 0), True, True, 4, 30),
    ("ragged_l2", (20, 12, 9), (0,) * 6, True, False, 2, 40),
]


def run_case(name, mx, bc, var, offd, L, max_outer):
    p = make_case(mx, bc, var, offd, salt=len(name))
    cfg = orc.make_cfg(p.mx, p.delx, p.bc, L=L, eps=1e-9, max_outer=max_outer, var_coef=var,
                       use_offdiag=offd, Dc=p.Dc)
    phi = p.phi.copy()
    cp = lambda a: None if a is None else a.copy()
    r = orc.solve(cfg, phi, p.rhs.copy(), p.lam.copy(), cp(p.Dxx), cp(p.Dyy), cp(p.Dzz), cp(p.Dxy), hashes=True)
    mid = (p.mx[2] // 2 + 1, p.mx[1] // 2 + 1)
    return {
        "name": name, "mx": list(p.mx), "bc": list(bc), "var": var, "offd": offd, "L": L,
        "max_outer": max_outer, "eps": 1e-9, "Dc": list(p.Dc),
        "status": r["status"], "iters": r["iters"],
        "norm_rhs": float(r["norm_rhs"]).hex(), "norm_init": float(r["norm_init"]).hex(),
        "history": [float(v).hex() for v in r["history"][:50]],
        "phi_sum": float(phi.sum()).hex(), "phi_abs_sum": float(np.abs(phi).sum()).hex(),
        "phi_centre_line": [float(v).hex() for v in phi[mid[0], mid[1], :]],
        # the reference's own hash() (XOR of the raw bits of every stored cell, solveLin_BICGStab.C:25-47) at its
        # disabled print sites: (phi, rhs, lambda) on entry (:284-287), (phi, residuals[0]) after every pass (:473-477)
        "input_hashes": [str(v) for v in r["input_hashes"]],
        "pass_hashes": [[str(a), str(b)] for a, b in r["pass_hashes"][:50]],
        "phi_hash": str(orc.hash_field(cfg, phi)),
    }


def main():
    orc.build()
    out = {"generator": "tests/golden/gen_golden.py", "oracle": "oracle/krylov_oracle.c",
           "pins": "the oracle (the reference has no golden vectors for this path)",
           "cases": [run_case(*c) for c in CASES]}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_cases.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    for c in out["cases"]:
        print(c["name"], "status", c["status"], "iters", c["iters"])


if __name__ == "__main__":
    main()
