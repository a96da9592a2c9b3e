#!/usr/bin/env python3
"""Kernel-tuning aid: run bench.py (one GPU, no CPU leg, no extras) under several settings of the
library's environment knobs and/or several builds of the library, one summary line each.
usage: tools/env_sweep.py OUT.jsonl 'bench args' [LIB=path] VAR=a,b,c [VAR2=...]   (cartesian product)"""
import itertools
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out, bargs = sys.argv[1], sys.argv[2].split()
axes = []
for spec in sys.argv[3:]:
    k, v = spec.split("=", 1)
    axes.append([(k, x) for x in v.split(",")])
with open(out, "w") as fo:
    for combo in itertools.product(*axes):
        env = dict(os.environ)
        for k, v in combo:
            if k == "LIB":
                if v != "default":
                    env["KRY_LIB_PATH"] = os.path.join(ROOT, v)
            elif v != "default":
                env[k] = v
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--no-also"] + bargs, env=env,
                           capture_output=True, text=True, timeout=600)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            line = {"set": dict(combo), "value": d["value"], "ms": d["ms_per_step"], "valid": d["valid"],
                    "kernels": {k: v["avg_ms"] for k, v in d["kernels"].items()}}
        except Exception:
            line = {"set": dict(combo), "error": r.stderr[-400:]}
        fo.write(json.dumps(line) + "\n")
        fo.flush()
        print(json.dumps(line), flush=True)
        if "error" in line:
            sys.exit(1)      # nothing more on the GPU after a failure
