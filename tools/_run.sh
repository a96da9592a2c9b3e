set -e
out=gpurun_out
timeout -k 10 400 python bench.py --ell 1 --no-also > $out/r3f_var512_l1_line.json 2> $out/r3f_extra.err
timeout -k 10 200 python bench.py --grid 1024 --weak --workload const --ell 1 --no-cpu --no-also > $out/r3f_weak_block_const_l1_line.json 2>> $out/r3f_extra.err
timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 3 > $out/r3f_bench512_two_ranks_one_gpu_line.json 2>> $out/r3f_extra.err
timeout -k 10 300 python -u tools/slab_shape.py > $out/r3f_slab_shapes.txt 2>> $out/r3f_extra.err
echo ok
