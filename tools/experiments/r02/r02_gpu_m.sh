#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_generate.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -15 > $O/r02m_tests.txt
cat $O/r02m_tests.txt
B15="--workload 1p5b --batch 8 --steps 100 --no-cpu-baseline --no-roofline"
timeout 300 python bench.py $B15 > $O/r02m_1p5b_b8_full.json 2>$O/r02m_err1.txt
VVHIP_BATCH_CODEC=heavy timeout 300 python bench.py $B15 > $O/r02m_1p5b_b8_heavy.json 2>/dev/null
B7="--batch 8 --speakers 4 --text-tokens 10569 --steps 20 --warmup 5 --skip-extra --no-cpu-baseline --no-roofline"
timeout 500 python bench.py $B7 > $O/r02m_7b_b8_full.json 2>$O/r02m_err2.txt
timeout 300 python bench.py --workload 1p5b --batch 4 --continuous 12 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/r02m_1p5b_continuous.json 2>/dev/null
for f in $O/r02m_*.json; do echo $(basename $f) $(python -c "import json;d=json.load(open('$f'));print(d['ms_per_step'], d['value'])"); done
tail -n 3 $O/r02m_err1.txt $O/r02m_err2.txt
