set -e
timeout -k 10 500 python -u -m pytest tests/test_gpu_bitwise.py tests/test_gpu_fuzz.py tests/test_gpu_resident.py tests/test_gpu_slabs.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r3a_tests.log 2>&1
timeout -k 10 200 python3 -u bench.py --grid 256 --workload const --ell 1 --steps 200 --warmup 20 --no-cpu --no-also > gpurun_out/r3a_const256_l1.json 2> gpurun_out/r3a.err
KRY_PROFILE=0 timeout -k 10 200 python3 -u bench.py --grid 256 --workload const --ell 1 --steps 200 --warmup 20 --no-cpu --no-also > gpurun_out/r3a_const256_l1_noprof.json 2>> gpurun_out/r3a.err
timeout -k 10 200 python3 -u bench.py --grid 256 --workload const --ell 2 --steps 100 --warmup 10 --no-cpu --no-also > gpurun_out/r3a_const256_l2.json 2>> gpurun_out/r3a.err
timeout -k 10 300 python3 -u bench.py --no-cpu --steps 20 > gpurun_out/r3a_bench512.json 2>> gpurun_out/r3a.err
echo ok
