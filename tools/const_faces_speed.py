#!/usr/bin/env python3
"""Constant coefficients on faces that are not all Dirichlet: passes/s of the fused general-boundary
kernels (the operator's input treatment evaluated inside them) against one kernel per primitive.
usage: tools/const_faces_speed.py [N]   (needs a GPU)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                    # noqa: E402
from krylov_amd import capi                     # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
f = bench.slab_problem(n, 1, 0, "const")
for bc in ((0, 0, 0, 0, 0, 0), (0, 0, 1, 1, 0, 1), (1, 1, 1, 1, 1, 1)):
    for L in (1, 2):
        for flags, name in ((0, "fused"), (capi.KRY_FORCE_GENERIC, "generic")):
            with capi.Solver(f["mx"], f["delx"], bc, L=L, eps=0.0, flags=flags) as s:
                s.load_const(f["phi"], f["rhs"], f["lam"], f["Dc"])
                s.iterate(5)
                t0 = time.perf_counter()
                s.iterate(40)
                dt = time.perf_counter() - t0
                print("%d^3 const l=%d faces %s %-8s path %d: %8.1f passes/s" % (n, L, bc, name, s.stats().fused, 40 / dt), flush=True)
