#!/usr/bin/env python3
"""Orchestration overhead of the slab decomposition, measured on ONE GPU: the
same 512^3 solve as P slabs of one process (device-to-device halo copies,
cross-slab sum, per-slab scalar kernels).  All slabs share the device, so the
ideal result is the single-slab rate."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from krylov_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
f = bench.slab_problem(n, 1, 0, "var")
for P in (1, 2, 4, 8):
    for ov in ("1", "0"):
        if P == 1 and ov == "0":
            continue
        os.environ["KRY_OVERLAP"] = ov
        s = capi.Solver(f["mx"], f["delx"], (0,) * 6, L=2, eps=0.0, nslabs=P, slab_count=P, devices=(0,))
        s.load_var(f["phi"], f["rhs"], f["lam"], f["Dxx"], f["Dyy"], f["Dzz"], None, False)
        s.iterate(3)
        t0 = time.perf_counter()
        s.iterate(20)
        dt = time.perf_counter() - t0
        st = s.stats()
        print("slabs %d overlap %s : %.2f it/s  %.3f ms/pass   host enqueue %.3f ms/pass (%.3f per slab)"
              % (P, ov, 20 / dt, dt / 20 * 1e3, st.enqueue_ms / 23, st.enqueue_ms / 23 / P), flush=True)
        s.close()
