#!/bin/bash
# 256^3 constant-coefficient profiles (kernel statistics, HBM traffic, SQ / L2 counters), l = 1 and 2.
# usage (on the GPU box, from the repository root): tools/r3_profile_const256.sh TAG
set -e
tag=$1
root=$PWD
out=$root/gpurun_out
mkdir -p $out
for L in 1 2; do
  tools/collect_profiles.sh ${tag}_const256_l$L const --grid 256 --ell $L --steps 40 --warmup 5
  cd /tmp
  args="--steps 12 --warmup 2 --no-cpu --no-also --workload const --grid 256 --ell $L"
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/prof_${tag}_const256_l${L}_sq -o p -- python3 $root/bench.py $args > /dev/null 2> $out/prof_${tag}_const256_l${L}_sq.err
  timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/prof_${tag}_const256_l${L}_tcc -o p -- python3 $root/bench.py $args > /dev/null 2> $out/prof_${tag}_const256_l${L}_tcc.err
  cd $root
done
echo done
