#!/bin/bash
# session F: 2-row GEMV form A/B
mkdir -p gpurun_out
O=gpurun_out
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40) > $O/r02f_pytest.log 2>&1
NS="--skip-extra --no-cpu-baseline --no-roofline --steps 40 --warmup 5"
timeout 300 python bench.py $NS > $O/r02f_ns_mr2.json 2>/dev/null
VVHIP_NO_MR2=1 timeout 300 python bench.py $NS > $O/r02f_ns_nomr2.json 2>/dev/null
timeout 300 python bench.py $NS > $O/r02f_ns_mr2_b.json 2>/dev/null
T15="--workload 1p5b --steps 150 --warmup 10 --no-cpu-baseline --no-roofline"
timeout 300 python bench.py $T15 > $O/r02f_1p5b_mr2.json 2>/dev/null
VVHIP_NO_MR2=1 timeout 300 python bench.py $T15 > $O/r02f_1p5b_nomr2.json 2>/dev/null
for f in $O/r02f_*.json; do echo $(basename $f) $(python -c "import json;d=json.load(open('$f'));print(d['ms_per_step'])"); done
grep -E "passed|failed" $O/r02f_pytest.log
