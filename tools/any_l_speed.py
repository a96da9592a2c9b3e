#!/usr/bin/env python3
"""BiCGStab(l) for l = 1..8: passes/s and operator applications/s of the fused kernels against one
kernel per primitive, on the benchmark's field-coefficient problem.
usage: tools/any_l_speed.py [N] [workload]   (needs a GPU)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                    # noqa: E402
from krylov_amd import capi                     # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kind = sys.argv[2] if len(sys.argv) > 2 else "var"
f = bench.slab_problem(n, 1, 0, kind)
for L in (1, 2, 3, 4, 6, 8):
    for flags, name in ((0, "fused"), (capi.KRY_FORCE_GENERIC, "generic")):
        with capi.Solver(f["mx"], f["delx"], f["bc"], L=L, eps=0.0, flags=flags) as s:
            if kind == "const":
                s.load_const(f["phi"], f["rhs"], f["lam"], f["Dc"])
            else:
                s.load_var(f["phi"], f["rhs"], f["lam"], f["Dxx"], f["Dyy"], f["Dzz"], f.get("Dxy"))
            s.iterate(4)
            reps = max(4, 40 // L)
            t0 = time.perf_counter()
            s.iterate(reps)
            dt = time.perf_counter() - t0
            st = s.stats()
            print("%d^3 %s l=%d %-8s path %d: %8.1f passes/s %8.1f operator applications/s  %.2f TB/s by the word count"
                  % (n, kind, L, name, st.fused, reps / dt, 2 * L * reps / dt, st.bytes_per_iter * reps / dt / 1e12), flush=True)
