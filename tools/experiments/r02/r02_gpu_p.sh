#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_timed_mode.py tests/test_gpu_kernels.py tests/test_gpu_generate.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -12 > $O/r02p_tests.txt
cat $O/r02p_tests.txt
B7="--batch 8 --speakers 4 --text-tokens 10569 --steps 20 --warmup 5 --skip-extra --no-cpu-baseline --no-roofline"
timeout 500 python bench.py $B7 > $O/r02p_7b_b8_p16.json 2>$O/r02p_err2.txt
VVHIP_P16_NO_OPROJ=1 timeout 500 python bench.py $B7 > $O/r02p_7b_b8_p16_noo.json 2>/dev/null
VVHIP_NO_P16=1 timeout 500 python bench.py $B7 > $O/r02p_7b_b8_old.json 2>/dev/null
B15="--workload 1p5b --batch 8 --steps 100 --no-cpu-baseline --no-roofline"
timeout 300 python bench.py $B15 > $O/r02p_1p5b_b8_p16.json 2>/dev/null
VVHIP_NO_P16=1 timeout 300 python bench.py $B15 > $O/r02p_1p5b_b8_old.json 2>/dev/null
timeout 300 python bench.py --workload 1p5b --batch 4 --continuous 12 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/r02p_1p5b_continuous.json 2>/dev/null
for f in $O/r02p_*.json; do echo $(basename $f) $(python -c "import json;d=json.load(open('$f'));print(d['ms_per_step'], d['value'])"); done
tail -n 3 $O/r02p_err2.txt
