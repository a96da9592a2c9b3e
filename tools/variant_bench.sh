#!/bin/bash
# usage: tools/variant_bench.sh OUT.jsonl "bench args" lib1.so lib2.so ...   (kernel tuning: one bench line per library build)
out=$1; args=$2; shift 2
: > $out
for lib in "$@"; do
  echo "{\"lib\": \"$lib\"}" >> $out
  KRY_LIB_PATH=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu $args >> $out 2>> $out.err || exit 1
done
