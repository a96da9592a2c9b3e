#!/bin/bash
# usage: tools/sweep_zc.sh WORKLOAD GRID "ZC LIST" OUTFILE  -- passes/s of bench.py per chunk depth
wl=$1; n=$2; out=$4
for zc in $3; do
  v=$(KRY_ZC=$zc KRY_PROFILE=0 timeout -k 10 120 python bench.py --grid $n --workload $wl --no-cpu --steps 30 --warmup 4 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["valid"])')
  echo "$wl $n zc=$zc $v" >> $out
done
