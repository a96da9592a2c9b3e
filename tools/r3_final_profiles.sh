#!/bin/bash
# Round-3 final tree: bench lines, rocprofv3 kernel statistics and HBM traffic per workload.
# usage (on the GPU box, from the repository root): tools/r3_final_profiles.sh
# leaves everything under gpurun_out/ (prof_r3f_* directories and r3f_*.json lines).
set -e
root=$PWD
out=$root/gpurun_out
mkdir -p $out
timeout -k 10 400 python bench.py > $out/r3f_bench512_line.json 2> $out/r3f_bench512.err
timeout -k 10 200 python bench.py --grid 256 --workload mixed --no-cpu --no-also > $out/r3f_mixed256_line.json 2> $out/r3f_mixed.err
timeout -k 10 200 python bench.py --grid 512 --workload mixed --no-cpu --no-also > $out/r3f_mixed512_line.json 2>> $out/r3f_mixed.err
timeout -k 10 200 python bench.py --grid 256 --workload mixed --ell 1 --no-cpu --no-also > $out/r3f_mixed256_l1_line.json 2>> $out/r3f_mixed.err
timeout -k 10 200 python bench.py --grid 512 --workload mixed --ell 1 --no-cpu --no-also > $out/r3f_mixed512_l1_line.json 2>> $out/r3f_mixed.err
timeout -k 10 200 python bench.py --grid 512 --workload const --no-cpu --no-also > $out/r3f_const512_l2_line.json 2>> $out/r3f_mixed.err
timeout -k 10 200 python bench.py --grid 1024 --weak --no-cpu --no-also > $out/r3f_weak_block_var_l2_line.json 2>> $out/r3f_mixed.err
tools/collect_profiles.sh r3f_var var
tools/collect_profiles.sh r3f_var_l1 var --ell 1
tools/collect_profiles.sh r3f_const256_l1 const --grid 256 --ell 1 --steps 40 --warmup 5
tools/collect_profiles.sh r3f_const256_l2 const --grid 256 --ell 2 --steps 40 --warmup 5
tools/collect_profiles.sh r3f_mixed256 mixed --grid 256
tools/collect_profiles.sh r3f_mixed512 mixed --grid 512
echo done
