#!/bin/bash
# The round's bench lines and profiles in one go (on the GPU box, from the repository root):
#   tools/round_measurements.sh TAG
# leaves gpurun_out/TAG_*.json and gpurun_out/prof_TAG_* ; every step under its own time limit, nothing after a failure.
set -e
tag=$1
out=gpurun_out
mkdir -p $out
timeout -k 10 400 python3 -u bench.py > $out/${tag}_bench512_line.json 2> $out/${tag}_bench512.err
timeout -k 10 200 python3 -u bench.py --weak --grid 1024 --no-cpu --no-also --steps 30 > $out/${tag}_weak_block_var_l2_line.json 2> $out/${tag}_weak.err
timeout -k 10 200 python3 -u bench.py --weak --grid 1024 --no-cpu --no-also --steps 30 --workload const --ell 1 > $out/${tag}_weak_block_const_l1_line.json 2>> $out/${tag}_weak.err
timeout -k 10 200 python3 -u bench.py --workload const --no-cpu --no-also --steps 40 > $out/${tag}_const512_l2_line.json 2> $out/${tag}_const.err
timeout -k 10 200 python3 -u bench.py --workload mixed --no-cpu --no-also --steps 40 > $out/${tag}_mixed512_line.json 2> $out/${tag}_mixed.err
timeout -k 10 200 python3 -u bench.py --workload mixed --grid 256 --no-cpu --no-also --steps 200 --warmup 20 > $out/${tag}_mixed256_line.json 2>> $out/${tag}_mixed.err
timeout -k 10 300 python3 -u bench.py --gpus 2 --no-cpu --steps 30 > $out/${tag}_bench512_two_ranks_one_gpu_line.json 2> $out/${tag}_two_ranks.err
tools/collect_profiles.sh ${tag}_var var
tools/collect_profiles.sh ${tag}_mixed256 mixed --grid 256
echo done
