# scratch driver for gpurun calls: put the commands of the next GPU run here
set -e
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > gpurun_out/s25_tests.log 2>&1
timeout -k 10 300 python bench.py > gpurun_out/s25_bench.json 2> gpurun_out/s25_bench.err
timeout -k 10 300 python -u tools/size_sweep_speed.py 32 64 80 128 256 > gpurun_out/s25_sizes.log 2>&1
echo ok
