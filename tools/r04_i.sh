#!/bin/bash
# round 4, pass i: short-prompt GEMM (K split + reduce, two stage buffers): its parity tests, then first-audio A/B of the stage buffers
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04i; mkdir -p $O
cd $R; export TMPDIR=/tmp
(time timeout 600 python -m pytest tests/test_gpu_shipped.py -k "short_prompt_gemm3 or gemm4_k_split" tests/test_gpu_kernels.py -m gpu -q) > $O/pytest_g3.log 2>&1; tail -4 $O/pytest_g3.log
Q="--no-cpu-baseline --no-eager-baseline --skip-extra --no-roofline"
for i in 1 2; do for k in 0 1; do
  VVHIP_G3_DB=$k timeout 200 python bench.py --workload 1p5b --steps 30 --warmup 5 $Q > $O/ab_db_1p5b_${k}_$i.json 2>/dev/null
  VVHIP_G3_DB=$k timeout 200 python bench.py --model 7b --workload 1p5b --solver-steps 10 --steps 30 --warmup 5 $Q > $O/ab_db_7bshort_${k}_$i.json 2>/dev/null
done; done
for f in $O/ab_db_*.json; do echo $(basename $f) $(python -c "
import json;d=json.load(open('$f'));print(d['ms_per_step'], d['extra']['prefill_phases']['lm_passes_s'], d['extra']['first_audio']['p50_ms'])"); done
