#!/bin/bash
# prefill A/B in ONE box: the library at HEAD vs the same library with round 3's prefill.hip (build/variants/libvvhip_prefill_r03.so)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04h; mkdir -p $O
cd $R; export TMPDIR=/tmp
Q="--no-cpu-baseline --no-eager-baseline --skip-extra --no-roofline --steps 10 --warmup 3"
for i in 1 2; do
  for v in head r03; do
    if [ $v = r03 ]; then export VVHIP_LIB=$R/build/variants/libvvhip_prefill_r03.so; else unset VVHIP_LIB; fi
    timeout 200 python bench.py $Q > $O/pf_${v}_$i.json 2>/dev/null
    echo $v $i $(python -c "
import json;d=json.load(open('$O/pf_${v}_$i.json'));print(d['ms_per_step'], d['extra']['prefill_phases']['lm_passes_s'], d['extra']['first_audio']['p50_ms'])")
  done
done
