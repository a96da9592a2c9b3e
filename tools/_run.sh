set -e
timeout -k 10 500 python -u -m pytest tests/test_gpu_bitwise.py tests/test_gpu_ops.py tests/test_gpu_fuzz.py tests/test_gpu_slabs.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r3f_tests.log 2>&1
out=gpurun_out/r3f.jsonl; : > $out
run() { echo "{\"variant\": \"$1\"}" >> $out; shift; env "$@" timeout -k 10 120 python3 -u bench.py $B >> $out 2>> gpurun_out/r3f.err; }
B="--grid 256 --workload const --ell 1 --steps 200 --warmup 20 --no-cpu --no-also"
for alt in 0 1; do for keep in 0 10 20 30 50; do
run c256l1_alt${alt}_keep$keep KRY_PROFILE=1 KRY_ALT=$alt KRY_KEEP_PCT=$keep
done; done
run c256l1_alt1_keep0_np KRY_PROFILE=0 KRY_ALT=1 KRY_KEEP_PCT=0
run c256l1_alt1_keep20_np KRY_PROFILE=0 KRY_ALT=1 KRY_KEEP_PCT=20
run c256l1_alt1_keep100 KRY_PROFILE=1 KRY_ALT=1 KRY_KEEP_PCT=100
B="--grid 256 --workload const --ell 2 --steps 100 --warmup 20 --no-cpu --no-also"
for keep in 0 20 40; do
run c256l2_alt1_keep$keep KRY_PROFILE=1 KRY_ALT=1 KRY_KEEP_PCT=$keep
done
run c256l2_alt0_keep0 KRY_PROFILE=1 KRY_ALT=0 KRY_KEEP_PCT=0
echo ok
