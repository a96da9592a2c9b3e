#!/bin/bash
# passes/s against the chunk depth of the stencil kernels (KRY_ZC), one bench line per value
# usage: tools/zc_sweep.sh OUT.jsonl "bench args" zc1 zc2 ...
out=$1; args=$2; shift 2
: > $out
for zc in "$@"; do
  echo "{\"zc\": $zc}" >> $out
  KRY_ZC=$zc timeout -k 10 200 python3 bench.py --no-cpu --no-also $args >> $out 2>> $out.err || exit 1
done
