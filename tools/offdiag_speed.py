#!/usr/bin/env python3
"""Field coefficients with the off-diagonal term (Dxy != 0, the 11-point operator): passes/s of the fused
kernels on Dirichlet and on mixed faces.  usage: tools/offdiag_speed.py [N]   (needs a GPU)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                              # noqa: E402
import bench                                    # noqa: E402
from krylov_amd import capi                     # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
f = bench.slab_problem(n, 1, 0, "var")
z, y, x = np.meshgrid(*(np.arange(-1, n + 2) / n,) * 3, indexing="ij", sparse=True)
dxy = np.ascontiguousarray(np.broadcast_to(0.1 * np.sin(2 * np.pi * x) * np.sin(2 * np.pi * y) * (1 + z), f["phi"].shape))
os.environ["KRY_PROFILE"] = "0"
for bc in ((0, 0, 0, 0, 0, 0), (0, 0, 1, 1, 0, 1), (1, 1, 1, 1, 1, 1)):
    for L in (1, 2):
        with capi.Solver(f["mx"], f["delx"], bc, L=L, eps=0.0) as s:
            s.load_var(f["phi"], f["rhs"], f["lam"], f["Dxx"], f["Dyy"], f["Dzz"], dxy, True)
            s.iterate(5)
            t0 = time.perf_counter()
            s.iterate(40)
            dt = time.perf_counter() - t0
            print("%d^3 var+Dxy l=%d faces %s path %d: %8.1f passes/s" % (n, L, bc, s.stats().fused, 40 / dt), flush=True)
