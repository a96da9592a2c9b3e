#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: prints what a build of the reference would print with its own disabled hash
prints switched on (BICGStab3D/solveLin_BICGStab.C:284-287 and the `#if 0` block at :472-477), for the
64^3 plumbing case of the reference driver (test 1: D = (1,1,10), lambda = 0.2 x y^2 z, Dirichlet faces,
l = 2, eps 1e-6; BICGStab3D/main.C:223-229, 277, 282-284, 456-463) or another size / test.

  python oracle/print_hashes.py [--size 64] [--ell 2] [--poisson]

The reference's hash() XORs the raw bit patterns of every stored cell of a field (:25-47), so two
runs print the same numbers only if every cell agrees bit for bit, ghost cells included.  Anyone who
can build the reference (with its real matrix.H) can enable those prints and diff the two outputs.
Line formats: the input block as at :284-287; per pass first the reference's own line (:475-477 --
it converts the 64-bit value to double and prints it with the stream's default 6 digits), then the
exact integers, which need `uint64_t` instead of `const double` at :473-474 on the reference's side."""
import argparse
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
sys.path[:] = [q for q in sys.path if os.path.abspath(q or ".") != _here]   # `oracle` is the directory, not oracle.py
sys.path.insert(0, os.path.dirname(_here))
from krylov_amd import problems as P   # (closed-form fields of the reference driver's test cases; no device code)
from oracle import oracle as orc


def sequences(n=64, L=2, poisson=False, eps=1e-6):
    p = P.poisson(n) if poisson else P.reference_test1(n)
    cfg = orc.make_cfg(p.mx, p.delx, p.bc, L=L, eps=eps, var_coef=False, Dc=p.Dc)
    phi = p.phi.copy()
    r = orc.solve(cfg, phi, p.rhs.copy(), p.lam.copy(), hashes=True)
    return r, orc.hash_field(cfg, phi)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--ell", type=int, default=2)
    ap.add_argument("--poisson", action="store_true", help="D = (1,1,1), lambda = 0, rhs = -3 pi^2 sin sin sin")
    a = ap.parse_args()
    r, final = sequences(a.size, a.ell, a.poisson)
    print("Input variables:")
    print("\thash(phi)     = %d" % r["input_hashes"][0])
    print("\thash(rhs)     = %d" % r["input_hashes"][1])
    print("\thash(lambda)  = %d" % r["input_hashes"][2])
    for k, (hp, hr) in enumerate(r["pass_hashes"]):
        print(" hash(phi) = %g, hash(residual) %g" % (float(hp), float(hr)))
        print(" [exact, pass %d] hash(phi) = %d, hash(residual) = %d" % (k + 1, hp, hr))
    print(" passes: %d, status %d, hash of the returned phi = %d" % (r["iters"], r["status"], final))


if __name__ == "__main__":
    main()
