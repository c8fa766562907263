#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04l; mkdir -p $O
cd $R; export TMPDIR=/tmp
Q="--no-cpu-baseline --no-eager-baseline --skip-extra --no-roofline"
for i in 1 2; do
  timeout 200 python bench.py --workload 1p5b --steps 30 --warmup 5 $Q > $O/pk_1p5b_$i.json 2>/dev/null
  timeout 200 python bench.py --model 7b --workload 1p5b --solver-steps 10 --steps 30 --warmup 5 $Q > $O/pk_7bshort_$i.json 2>/dev/null
done
timeout 200 python bench.py --steps 10 --warmup 3 $Q > $O/pk_7b.json 2>/dev/null
for f in $O/pk_*.json; do echo $(basename $f) $(python -c "
import json;d=json.load(open('$f'));print(d['ms_per_step'], d['extra']['prefill_phases']['lm_passes_s'], d['extra']['first_audio']['p50_ms'])"); done
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "prefill or pack" 2>&1 | tail -2
