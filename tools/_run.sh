# scratch driver for gpurun calls: put the commands of the next GPU run here
set -e
timeout -k 10 600 python -u -m pytest tests/test_gpu_bitwise.py -k "constant_coefficients" -x -q --timeout 180 --timeout-method thread
root=$PWD; export TMPDIR=/tmp; cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/s37_stats -o p -- python3 $root/tools/const_faces_kernels.py 256 2
