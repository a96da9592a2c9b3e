#!/bin/bash
# Bench lines of the other configurations, each with its CPU leg where that is cheap (on the GPU box, from the repository root).
# usage: tools/extra_lines.sh TAG ; a line that reports "valid": false ends the script
set -e
tag=$1
out=gpurun_out
mkdir -p $out
timeout -k 10 250 python3 -u bench.py --workload mixed --grid 256 --no-also --steps 60 --warmup 10 > $out/${tag}_mixed256_line.json 2> $out/${tag}_extra.err
timeout -k 10 250 python3 -u bench.py --workload mixed --no-cpu --no-also > $out/${tag}_mixed512_line.json 2>> $out/${tag}_extra.err
timeout -k 10 250 python3 -u bench.py --workload const --grid 256 --ell 1 --no-also --steps 200 --warmup 20 > $out/${tag}_const256_l1_line.json 2>> $out/${tag}_extra.err
timeout -k 10 250 python3 -u bench.py --workload const --grid 256 --ell 2 --no-also --steps 100 --warmup 10 > $out/${tag}_const256_l2_line.json 2>> $out/${tag}_extra.err
timeout -k 10 250 python3 -u bench.py --ell 1 --no-also > $out/${tag}_var512_l1_line.json 2>> $out/${tag}_extra.err
echo done
