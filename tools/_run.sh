set -e
timeout -k 10 500 python3 -u bench.py > gpurun_out/r3k_bench512.json 2> gpurun_out/r3k.err
timeout -k 10 300 python3 -u bench.py --gpus 2 --no-cpu --steps 20 > gpurun_out/r3k_two_ranks.json 2>> gpurun_out/r3k.err
echo ok
