#!/usr/bin/env python3
"""Per-pass distance between the device's residual history and the oracle's in the reference's
summation order, |norm_dev - norm_ref| / normRHS, and the distance of the converged solutions,
for the 64^3 constant-coefficient plumbing case (l = 1, 2) and a 32^3 field-coefficient case.
Needs a GPU.  usage: tools/history_drift.py OUT.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from krylov_amd import capi, problems as P      # noqa: E402
from oracle import oracle as orc                # noqa: E402
from helpers import oracle_solve, points, rel_l2  # noqa: E402

orc.build()
out = {"what": "device history vs oracle in the reference's summation order; drift[k] = |norm_dev - norm_ref| / normRHS at pass k+1",
       "eps": 1e-12, "cases": []}
for name, p, L in (("64^3 reference test 1 (constant coefficients), l=1", P.reference_test1(64), 1),
                   ("64^3 reference test 1 (constant coefficients), l=2", P.reference_test1(64), 2),
                   ("32^3 smooth field coefficients, l=2", P.smooth_var(32), 2)):
    want = oracle_solve(orc, p, L=L, eps=1e-12)
    got = capi.solve_problem(p, L=L, eps=1e-12)
    n = min(len(want["history"]), len(got["history"]))
    drift = np.abs(got["history"][:n] - want["history"][:n]) / want["norm_rhs"]
    out["cases"].append({"case": name, "passes_device": int(got["iters"]), "passes_oracle": int(want["iters"]),
                         "solution_rel_l2": float(rel_l2(points(got["phi"]), points(want["phi"]))),
                         "max_drift": float(drift.max()), "max_drift_first_50": float(drift[:50].max()),
                         "relative_history_oracle": [float(v / want["norm_rhs"]) for v in want["history"][:n]],
                         "drift": [float(v) for v in drift]})
json.dump(out, open(sys.argv[1], "w"), indent=1)
for c in out["cases"]:
    print(c["case"], c["passes_device"], c["passes_oracle"], c["solution_rel_l2"], c["max_drift"])
