# scratch driver for gpurun calls: put the commands of the next GPU run here
set -e
KRY_LIB_PATH=$PWD/krylov_amd/variants/libkrylov_amd_noinl.so timeout -k 10 300 python -u tools/const_faces_profile.py 256 2 > gpurun_out/s22_noinl.log 2>&1
KRY_LIB_PATH=$PWD/krylov_amd/variants/libkrylov_amd_noinl.so timeout -k 10 300 python -u -m pytest tests/test_gpu_bitwise.py -x -q -k "constant_coefficients" --timeout 240 --timeout-method thread > gpurun_out/s22_tests.log 2>&1
echo ok
