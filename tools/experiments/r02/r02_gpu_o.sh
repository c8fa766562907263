#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_timed_mode.py -x -q -m gpu 2>&1 | grep -v "^$" | tail -8 > $O/r02o_tests.txt
cat $O/r02o_tests.txt
NS="--skip-extra --no-cpu-baseline --no-roofline --steps 20 --warmup 5"
timeout 400 python bench.py $NS > $O/r02o_ns_new.json 2>$O/r02o_err1.txt
VVHIP_NO_ADA_GEMM3=1 timeout 400 python bench.py $NS > $O/r02o_ns_old.json 2>/dev/null
B7="--batch 8 --speakers 4 --text-tokens 10569 --steps 20 --warmup 5 --skip-extra --no-cpu-baseline --no-roofline"
timeout 500 python bench.py $B7 > $O/r02o_7b_b8_new.json 2>$O/r02o_err2.txt
B15="--workload 1p5b --batch 8 --steps 100 --no-cpu-baseline --no-roofline"
timeout 300 python bench.py $B15 > $O/r02o_1p5b_b8_new.json 2>/dev/null
for f in $O/r02o_*.json; do echo $(basename $f) $(python -c "import json;d=json.load(open('$f'));print(d['ms_per_step'], d['value'])"); done
tail -n 3 $O/r02o_err1.txt $O/r02o_err2.txt
