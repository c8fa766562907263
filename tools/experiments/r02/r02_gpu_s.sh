#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out
NS="--skip-extra --no-cpu-baseline --no-roofline --steps 10 --warmup 3"
timeout 400 python bench.py $NS > $O/r02s_ns_g4.json 2>$O/r02s_err1.txt
VVHIP_GEMM4=0 timeout 400 python bench.py $NS > $O/r02s_ns_g3.json 2>/dev/null
B7="--batch 8 --speakers 4 --text-tokens 10569 --steps 20 --warmup 5 --skip-extra --no-cpu-baseline --no-roofline"
timeout 500 python bench.py $B7 > $O/r02s_7b_b8.json 2>$O/r02s_err2.txt
for f in $O/r02s_*.json; do echo $(basename $f) $(python -c "import json;d=json.load(open('$f'));e=d['extra'];print(d['ms_per_step'], d['value'], e.get('prefill_phases'), e.get('prefill_tflops'))"); done
tail -n 3 $O/r02s_err1.txt
