set -e
timeout -k 10 900 python -u -m pytest tests/test_gpu_multiproc.py -x -q --timeout 300 --timeout-method thread -k "mapped_memory_exchange_reproduces or cartesian" > gpurun_out/r3p_tests.log 2>&1
echo ok
