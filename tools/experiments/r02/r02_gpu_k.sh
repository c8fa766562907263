#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out
T15="--workload 1p5b --steps 150 --warmup 10 --no-cpu-baseline --no-roofline"
for th in 512 128 64 32 0; do
VVHIP_WIDE4_WGS=$th timeout 300 python bench.py $T15 > $O/r02k_1p5b_wide4_$th.json 2>/dev/null
done
NS="--skip-extra --no-cpu-baseline --no-roofline --steps 20 --warmup 5"
for th in 512 64; do VVHIP_WIDE4_WGS=$th timeout 300 python bench.py $NS > $O/r02k_ns_wide4_$th.json 2>/dev/null; done
for f in $O/r02k_*.json; do echo $(basename $f) $(python -c "import json;d=json.load(open('$f'));print(d['ms_per_step'])"); done
