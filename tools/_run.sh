set -e
timeout -k 10 800 python -u -m pytest tests/test_gpu_fuzz.py tests/test_gpu_bitwise.py tests/test_gpu_ops.py tests/test_gpu_slabs.py -x -q --timeout 180 --timeout-method thread > gpurun_out/s6_tests.log 2>&1
for g in 256 512; do
  timeout -k 10 200 python bench.py --grid $g --workload mixed --steps 30 --warmup 5 --no-cpu --no-also > gpurun_out/s6_mixed${g}.json 2> gpurun_out/s6_mixed${g}.err
done
timeout -k 10 200 python bench.py --grid 256 --workload mixed --ell 1 --steps 30 --warmup 5 --no-cpu --no-also > gpurun_out/s6_mixed256_l1.json 2>> gpurun_out/s6_mixed256.err
echo ok
