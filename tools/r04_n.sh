#!/bin/bash
# unit-packed prefill attention (4 (row tile, head) units per workgroup): parity tests, then A/B against the previous build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04n; mkdir -p $O
cd $R; export TMPDIR=/tmp
true
Q="--no-cpu-baseline --no-eager-baseline --skip-extra --no-roofline --no-parity"
for i in 1 2; do
  for v in prev new; do
    if [ $v = prev ]; then export VVHIP_LIB="$R/tools/experiments/ab/libvvhip_prev.so"; else unset VVHIP_LIB; fi
    timeout 200 python bench.py --steps 5 --warmup 2 $Q > $O/ab_7b_${v}_$i.json 2>$O/err_7b_${v}_$i.txt
    timeout 200 python bench.py --workload 1p5b --steps 10 --warmup 3 $Q > $O/ab_1p5b_${v}_$i.json 2>/dev/null
  done
done
for f in $O/ab_*.json; do echo $(basename $f) $(python -c "
import json;d=json.load(open('$f'));print(d['extra']['libvvhip_build_id'], d['ms_per_step'], d['extra']['prefill_phases']['lm_passes_s'], d['extra']['first_audio']['p50_ms'])"); done
