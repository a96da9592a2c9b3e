#!/usr/bin/env python3
"""Pass rate of one GPU on the block an N-GPU run of the 512^3 problem gives it
(512 x 512 x 512/N, alone, without neighbours): what the block shape by itself costs
against 1/N of the undivided solve."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from krylov_amd import capi

os.environ.setdefault("KRY_PROFILE", "0")
base = None
for P in (1, 2, 4, 8):
    f = bench.slab_problem(512, 1, 0, "var", nz=512 // P)
    with capi.Solver(f["mx"], f["delx"], f["bc"], L=2, eps=0.0) as s:
        s.load_var(f["phi"], f["rhs"], f["lam"], f["Dxx"], f["Dyy"], f["Dzz"], None, False)
        s.iterate(5)
        t0 = time.perf_counter()
        s.iterate(40)
        dt = (time.perf_counter() - t0) / 40
        zc = s.stats().chunk_planes
    base = base or dt
    print("512 x 512 x %3d : %.3f ms/pass, %.1f passes/s, chunk %d planes, %.1f %% of 1/%d of the undivided pass"
          % (512 // P, dt * 1e3, 1 / dt, zc, 100 * base / P / dt, P), flush=True)
