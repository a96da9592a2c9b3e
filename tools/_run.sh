set -e
timeout -k 10 500 python -u -m pytest tests/test_gpu_bitwise.py tests/test_gpu_ops.py tests/test_gpu_solve.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r3e_tests.log 2>&1
out=gpurun_out/r3e.jsonl; : > $out
run() { echo "{\"variant\": \"$1\"}" >> $out; shift; env "$@" timeout -k 10 120 python3 -u bench.py $B >> $out 2>> gpurun_out/r3e.err; }
B="--grid 256 --workload const --ell 1 --steps 200 --warmup 20 --no-cpu --no-also"
run c256l1 KRY_PROFILE=0
run c256l1_ev KRY_PROFILE=1
B="--grid 256 --workload const --ell 2 --steps 100 --warmup 20 --no-cpu --no-also"
run c256l2 KRY_PROFILE=0
run c256l2_ev KRY_PROFILE=1
B="--grid 512 --workload const --ell 1 --steps 40 --warmup 5 --no-cpu --no-also"
run c512l1_ev KRY_PROFILE=1
B="--grid 512 --workload const --ell 2 --steps 40 --warmup 5 --no-cpu --no-also"
run c512l2_ev KRY_PROFILE=1
B="--grid 256 --workload mixed --ell 2 --steps 100 --warmup 10 --no-cpu --no-also"
run m256l2_ev KRY_PROFILE=1
echo ok
