#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_timed_mode.py tests/test_gpu_kernels.py tests/test_gpu_generate.py tests/test_gpu_streaming.py tests/test_gpu_geometry.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -12 > $O/r02q_tests.txt
cat $O/r02q_tests.txt
