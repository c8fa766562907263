#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out
NS="--skip-extra --no-cpu-baseline --no-roofline --steps 3 --warmup 1"
VVHIP_TIME_PREFILL_DETAIL=1 timeout 600 python bench.py $NS > $O/r02v_ns.json 2> $O/r02v_err.txt
python -c "import json;d=json.load(open('$O/r02v_ns.json'));print(d['ms_per_step'], d['extra']['prefill_phases'])"
grep "prefill detail" $O/r02v_err.txt $O/r02v_ns.json | head
