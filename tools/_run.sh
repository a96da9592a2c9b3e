set -e
timeout -k 10 300 python -u -m pytest tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r3m_tests.log 2>&1
echo ok
