#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output) into the
per-kernel HBM traffic summary that bench.py reads (profiles/*_hbm_traffic.json).

usage: summarize_pmc.py FETCH_counter_collection.csv WRITE_counter_collection.csv "WORKLOAD" OUT.json
"""
import collections
import csv
import json
import sys

NAMES = {"k_fused<0": "fused_A0", "k_fused<2": "fused_B0", "k_fused<1": "fused_A1", "k_fused<3": "fused_B0L",
         "k_fused<4": "fused_B1L", "k_mgs2": "mgs2", "k_final<": "final", "k_fused<5": "fused_residual",
         "g_bc": "bc_faces", "k_surf<": "surface"}


def agg(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in d.items()}


def main():
    f, w = agg(sys.argv[1]), agg(sys.argv[2])
    out = {"command": "rocprofv3 --pmc FETCH_SIZE (and, in a second run, WRITE_SIZE) --output-format csv -- "
                      "python3 bench.py --steps 10 --warmup 2 --no-cpu",
           "workload": sys.argv[3],
           "note": "FETCH_SIZE and WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies 128-byte read requests of a "
                   "wide coalesced stream as 64 bytes, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact for "
                   "16-byte-per-lane streaming stores.",
           "kernels": {}}
    for k in f:
        key = [x for x in NAMES if x in k]
        if not key or k not in w:
            continue
        fe, wr = f[k][0] * 1024, w[k][0] * 1024
        out["kernels"][NAMES[key[0]]] = {
            "kernel": k, "launches_sampled": f[k][1], "FETCH_SIZE_KiB_avg": f[k][0], "WRITE_SIZE_KiB_avg": w[k][0],
            "hbm_read_bytes_per_launch": 2 * fe, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": 2 * fe + wr}
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    for n, v in out["kernels"].items():
        print("%-16s read %.3f GB  write %.3f GB" % (n, v["hbm_read_bytes_per_launch"] / 1e9, v["hbm_write_bytes_per_launch"] / 1e9))


if __name__ == "__main__":
    main()
