#!/usr/bin/env python3
"""Passes per second and fraction of the HBM peak (by the contract's words) on grids of the reference's sweep
protocol (n = 80 ... 276, no powers of two), field and constant coefficients, Dirichlet faces, l = 2.
usage: tools/size_sweep_speed.py [n ...]   (needs a GPU)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["KRY_PROFILE"] = "0"
import bench                                    # noqa: E402
from krylov_amd import capi                     # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [80, 100, 128, 132, 160, 200, 252, 256, 260, 276]
for n in sizes:
    for wl in ("var", "const"):
        f = bench.slab_problem(n, 1, 0, wl)
        with capi.Solver(f["mx"], f["delx"], f["bc"], L=2, eps=0.0) as s:
            if wl == "var":
                s.load_var(f["phi"], f["rhs"], f["lam"], f["Dxx"], f["Dyy"], f["Dzz"], None, False)
            else:
                s.load_const(f["phi"], f["rhs"], f["lam"], f["Dc"])
            s.iterate(10)
            steps = 100
            t0 = time.perf_counter()
            s.iterate(steps)
            dt = time.perf_counter() - t0
            st = s.stats()
            W = bench.contract_words(2, wl == "var")
            frac = 8.0 * W * (n + 1) ** 3 / (dt / steps) / 1e9 / bench.HBM_PEAK_GBS
            print("n = %3d %-5s l=2: %8.1f passes/s, %.3f ms per pass, %.3f of peak by the contract's words, chunk %d planes"
                  % (n, wl, steps / dt, dt / steps * 1e3, frac, st.chunk_planes), flush=True)
