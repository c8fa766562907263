#!/bin/bash
# no-refresh tests on the GPU + the N > 1 control flow of bench.py on the one-GPU box (VVHIP_BENCH_SHARED_GPU=1: gloo, every rank on cuda:0)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04m; mkdir -p $O
cd $R; export TMPDIR=/tmp
true
( time VVHIP_BENCH_SHARED_GPU=1 timeout 300 python bench.py --gpus 2 --workload 1p5b --steps 20 --warmup 3 ) > $O/shared_gpu_n2_1p5b.json 2> $O/shared_gpu_n2_1p5b.err
tail -5 $O/shared_gpu_n2_1p5b.err; head -c 600 $O/shared_gpu_n2_1p5b.json; echo
( time VVHIP_BENCH_SHARED_GPU=1 timeout 400 python bench.py --gpus 2 --steps 10 --warmup 2 ) > $O/shared_gpu_n2_default.json 2> $O/shared_gpu_n2_default.err
tail -5 $O/shared_gpu_n2_default.err; head -c 600 $O/shared_gpu_n2_default.json; echo
