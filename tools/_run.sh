# scratch driver for gpurun calls: put the commands of the next GPU run here
set -e
timeout -k 10 900 python bench.py --grid 1024 --no-cpu --no-also --steps 6 --warmup 2 > gpurun_out/r3f_whole1024_line.json 2> gpurun_out/r3f_whole1024.err
echo ok
