#!/usr/bin/env python3
"""Config 1 on the GPU: the 64^3 problems of the CPU baseline's `c1` leg solved in full (eps 1e-6),
host fields in, solution out.  Prints one JSON line.  usage: tools/c1_on_gpu.py   (needs a GPU)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("KRY_PROFILE", "0")
from krylov_amd import capi, problems as P      # noqa: E402

out = []
for name, p, L in (("64^3 reference driver test 1, l=2", P.reference_test1(64), 2), ("64^3 Poisson, l=1", P.poisson(64), 1)):
    capi.solve_problem(p, L=L, eps=1e-6)        # first call: library and device warm-up
    best = None
    for _ in range(5):
        t0 = time.perf_counter()
        r = capi.solve_problem(p, L=L, eps=1e-6)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out.append({"workload": name, "iterations": int(r["iters"]), "status": int(r["status"]),
                "seconds_host_to_host": round(best, 5), "device_loop_ms": round(float(r.get("solve_ms", float("nan"))), 4),
                "relative_residual": float(r["history"][-1] / r["norm_rhs"]) if len(r["history"]) else None})
print(json.dumps({"c1_on_gpu": out}))
