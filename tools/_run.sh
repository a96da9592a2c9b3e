# scratch driver for gpurun calls: put the commands of the next GPU run here
set -e
timeout -k 10 500 python -u tools/size_sweep_speed.py > gpurun_out/s15_sizes.log 2>&1
echo ok
