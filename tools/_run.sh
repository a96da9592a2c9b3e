set -e
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r3j_tests.log 2>&1
timeout -k 10 400 python3 -u bench.py --no-cpu > gpurun_out/r3j_bench512.json 2> gpurun_out/r3j.err
timeout -k 10 200 python3 -u tools/slab_shape.py > gpurun_out/r3j_slab_shapes.txt 2>> gpurun_out/r3j.err
timeout -k 10 200 python3 -u bench.py --gpus 2 --no-cpu --steps 20 > gpurun_out/r3j_two_ranks.json 2>> gpurun_out/r3j.err
echo ok
