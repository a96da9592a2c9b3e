#!/usr/bin/env python3
"""A constant-coefficient solve on mixed faces, fused path only, for a kernel trace:
rocprofv3 --kernel-trace --stats -- python3 tools/const_faces_kernels.py [N] [l] [var]   (needs a GPU)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["KRY_PROFILE"] = "0"
import bench                                    # noqa: E402
from krylov_amd import capi                     # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
wl = "var" if len(sys.argv) > 3 else "const"
f = bench.slab_problem(n, 1, 0, wl)
with capi.Solver(f["mx"], f["delx"], (0, 0, 1, 1, 0, 1), L=L, eps=0.0) as s:
    if wl == "var":
        s.load_var(f["phi"], f["rhs"], f["lam"], f["Dxx"], f["Dyy"], f["Dzz"], None, False)
    else:
        s.load_const(f["phi"], f["rhs"], f["lam"], f["Dc"])
    s.iterate(45)
    print(s.stats().iterations)
