#!/usr/bin/env python3
"""Generates tests/golden/c1_hashes.json: the reference's hash() (BICGStab3D/solveLin_BICGStab.C:25-47) of phi
and of residuals[0] after every pass of the 64^3 plumbing case -- the reference driver's test 1
(BICGStab3D/main.C:223-229, 277, 282-284: D = (1,1,10), lambda = 0.2 x y^2 z, phi_ana = sin sin sin,
Dirichlet faces, phi_0 = 0), l = 2, eps 1e-6 (main.C:16, 456-463) -- computed by the CPU oracle in the
reference's own summation order, plus the hashes of the inputs at the reference's print sites (:284-287).
Like every fixture here it pins the oracle; it is also what a build of the reference with those prints
enabled must print (oracle/print_hashes.py shows it in the reference's line format).

Run from the repository root:  python tests/golden/gen_c1_hashes.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402
from krylov_amd import problems as P  # noqa: E402


def run(n=64, L=2):
    p = P.reference_test1(n)
    cfg = orc.make_cfg(p.mx, p.delx, p.bc, L=L, eps=1e-6, var_coef=False, Dc=p.Dc)
    phi = p.phi.copy()
    r = orc.solve(cfg, phi, p.rhs.copy(), p.lam.copy(), hashes=True)
    return {"case": "reference driver test 1, %d^3, l = %d, eps 1e-6, Dirichlet faces, serial summation order" % (n, L),
            "iters": r["iters"], "status": r["status"],
            "input_hashes": {"phi": str(r["input_hashes"][0]), "rhs": str(r["input_hashes"][1]),
                             "lambda": str(r["input_hashes"][2])},
            "pass_hashes": [{"phi": str(a), "residual": str(b)} for a, b in r["pass_hashes"]],
            "returned_phi_hash": str(orc.hash_field(cfg, phi)),
            "history": [float(v).hex() for v in r["history"]]}


def main():
    orc.build()
    out = run()
    out["generator"] = "tests/golden/gen_c1_hashes.py"
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c1_hashes.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(out["case"], "passes", out["iters"], "status", out["status"])


if __name__ == "__main__":
    main()
