set -e
timeout -k 10 800 python -u -m pytest tests/test_gpu_fuzz.py tests/test_gpu_bitwise.py tests/test_gpu_ops.py tests/test_gpu_slabs.py tests/test_gpu_fullsize.py -x -q --timeout 180 --timeout-method thread > gpurun_out/s7_tests.log 2>&1
for g in 256 512; do
  timeout -k 10 200 python bench.py --grid $g --workload mixed --steps 30 --warmup 5 --no-cpu --no-also > gpurun_out/s7_mixed${g}.json 2> gpurun_out/s7_mixed${g}.err
  timeout -k 10 200 python bench.py --grid $g --workload mixed --ell 1 --steps 30 --warmup 5 --no-cpu --no-also > gpurun_out/s7_mixed${g}_l1.json 2>> gpurun_out/s7_mixed${g}.err
done
echo ok
