#!/bin/bash
# rocprofv3 kernel statistics and HBM traffic counters of bench.py, per workload.
# usage (on the GPU box, from the repository root): tools/collect_profiles.sh TAG WORKLOAD [more bench args]
# leaves gpurun_out/prof_TAG_{stats,fetch,write}/ ; every step under its own time limit, nothing after a failure.
set -e
tag=$1; wl=$2; shift 2
root=$PWD
out=$root/gpurun_out
export TMPDIR=/tmp
cd /tmp
args="--steps 12 --warmup 2 --no-cpu --no-also --workload $wl $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_stats -o p -- python3 $root/bench.py $args > $out/prof_${tag}_line.json 2> $out/prof_${tag}_stats.err
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/prof_${tag}_fetch -o p -- python3 $root/bench.py $args > /dev/null 2> $out/prof_${tag}_fetch.err
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/prof_${tag}_write -o p -- python3 $root/bench.py $args > /dev/null 2> $out/prof_${tag}_write.err
cd $root
find $out/prof_${tag}_stats $out/prof_${tag}_fetch $out/prof_${tag}_write -name "*.csv" | head -20
