set -e
timeout -k 10 500 python -u -m pytest tests/test_gpu_compat.py tests/test_gpu_resident.py -x -q --timeout 180 --timeout-method thread > gpurun_out/r3l_tests.log 2>&1
echo ok
