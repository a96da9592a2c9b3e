#!/bin/bash
# usage: tools/sweep_zc_kernels.sh WORKLOAD GRID "ZC LIST" OUTFILE -- per-kernel ms of bench.py per chunk depth
wl=$1; n=$2; out=$4
for zc in $3; do
  v=$(KRY_ZC=$zc timeout -k 10 120 python bench.py --grid $n --workload $wl --no-cpu --steps 20 --warmup 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], " ".join("%s=%.4f" % (k, v["avg_ms"]) for k, v in d["kernels"].items()))')
  echo "$wl $n zc=$zc $v" >> $out
done
