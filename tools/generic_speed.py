#!/usr/bin/env python3
"""Pass rate of the generic (one kernel per primitive) path against the fused
path, same problem, one GPU."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from krylov_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
f = bench.slab_problem(n, 1, 0, "var")
for name, bc, flags in (("fused, Dirichlet", (0,) * 6, 0), ("generic, Dirichlet", (0,) * 6, capi.KRY_FORCE_GENERIC),
                        ("generic, mixed faces", (0, 0, 1, 1, 0, 1), 0)):
    for L in (1, 2):
        s = capi.Solver(f["mx"], f["delx"], bc, L=L, eps=0.0, flags=flags)
        s.load_var(f["phi"], f["rhs"], f["lam"], f["Dxx"], f["Dyy"], f["Dzz"], None, False)
        s.iterate(3)
        t0 = time.perf_counter()
        s.iterate(20)
        dt = time.perf_counter() - t0
        print("%-22s l=%d n=%d : %8.2f it/s  %.3f ms/pass" % (name, L, n, 20 / dt, dt / 20 * 1e3), flush=True)
        s.close()
