#!/usr/bin/env python3
"""Per-kernel times of the constant-coefficient solve on mixed faces (fused general-boundary kernels with the
operator's input treatment).  usage: tools/const_faces_profile.py [N] [l]   (needs a GPU)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["KRY_PROFILE"] = "1"
import bench                                    # noqa: E402
from krylov_amd import capi                     # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = bench.slab_problem(n, 1, 0, "const")
for bc in ((0, 0, 0, 0, 0, 0), (0, 0, 1, 1, 0, 1)):
    with capi.Solver(f["mx"], f["delx"], bc, L=L, eps=0.0) as s:
        s.load_const(f["phi"], f["rhs"], f["lam"], f["Dc"])
        s.iterate(5)
        a = s.stats()
        s.iterate(40)
        b = s.stats()
        print(bc, "chunk planes", b.chunk_planes, "tile rows", b.tile_rows,
              {bench.KIND_NAMES[k]: round((b.kernel_ms[k] - a.kernel_ms[k]) / max(1, b.kernel_calls[k] - a.kernel_calls[k]), 4)
               for k in range(8) if b.kernel_calls[k] > a.kernel_calls[k]}, flush=True)
