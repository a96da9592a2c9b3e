#!/usr/bin/env python3
"""Two slabs of one process on ONE device: passes per second and host time to queue a pass with the
slabs exchanging by copies and events (KRY_SLAB_LINK=0) and through each other's device memory
(KRY_SLAB_LINK=1).  usage: slab_link_speed.py [grid] [l] [passes]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from krylov_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
f = bench.slab_problem(n, 1, 0, "var")
os.environ["KRY_PROFILE"] = "0"
for nsl in (1, 2):
    for link in (("0",) if nsl == 1 else ("0", "1")):
        os.environ["KRY_SLAB_LINK"] = link
        with capi.Solver(f["mx"], f["delx"], f["bc"], L=L, eps=0.0, max_outer=0, nslabs=nsl, slab_begin=0,
                         slab_count=nsl, devices=(0,)) as s:
            s.load_var(f["phi"], f["rhs"], f["lam"], f["Dxx"], f["Dyy"], f["Dzz"], None, False)
            s.iterate(10)
            e0 = s.stats().enqueue_ms
            t0 = time.perf_counter()
            s.iterate(steps)
            dt = time.perf_counter() - t0
            st = s.stats()
            print("%d^3 var l=%d, %d slab(s), link %s (%s): %.1f passes/s, %.3f ms per pass, host queues a pass in %.3f ms, "
                  "last residual %.6e" % (n, L, nsl, st.slab_link, s.slab_link_note(), steps / dt, dt / steps * 1e3,
                                          (st.enqueue_ms - e0) / steps, s.history()[-1]), flush=True)
