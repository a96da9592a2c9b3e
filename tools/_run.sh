# scratch driver for gpurun calls: put the commands of the next GPU run here
set -e
timeout -k 10 300 python bench.py --no-cpu > gpurun_out/s27_bench.json 2> gpurun_out/s27.err
timeout -k 10 300 python bench.py --no-cpu > gpurun_out/s27_bench2.json 2>> gpurun_out/s27.err
echo ok
