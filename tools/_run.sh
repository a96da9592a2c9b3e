set -e
timeout -k 10 800 python -u -m pytest tests/test_gpu_fuzz.py tests/test_gpu_bitwise.py tests/test_gpu_ops.py tests/test_gpu_fullsize.py -x -q --timeout 180 --timeout-method thread > gpurun_out/s10_tests.log 2>&1
echo ok
