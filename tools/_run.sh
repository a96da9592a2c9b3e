# scratch driver for gpurun calls: put the commands of the next GPU run here
set -e
timeout -k 10 900 python -u -m pytest tests/test_gpu_fuzz.py tests/test_gpu_bitwise.py tests/test_gpu_ops.py tests/test_gpu_slabs.py tests/test_gpu_solve.py -x -q --timeout 240 --timeout-method thread > gpurun_out/s16_tests.log 2>&1
timeout -k 10 500 python -u tools/size_sweep_speed.py > gpurun_out/s16_sizes.log 2>&1
echo ok
