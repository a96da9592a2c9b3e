#!/bin/bash
# usage: tools/build_variant.sh NAME "-DFLAG ..."   -> krylov_amd/variants/libkrylov_amd_NAME.so (kernel tuning experiments)
set -e
name=$1; flags=$2
cd krylov_amd/csrc
mkdir -p ../variants
make -s kry_solver.o kry_comm.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-value -Wno-unused-result $flags -c kry_kernels.hip -o /tmp/kry_kernels_$name.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../variants/libkrylov_amd_$name.so /tmp/kry_kernels_$name.o kry_solver.o kry_comm.o -ldl
echo built krylov_amd/variants/libkrylov_amd_$name.so
