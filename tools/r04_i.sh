#!/bin/bash
# round 4, pass i: K-split of the short-prompt GEMM (vv_gemm3 + vv_g3_reduce): parity tests, then first-audio A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04i; mkdir -p $O
cd $R; export TMPDIR=/tmp
(time timeout 900 python -m pytest tests/test_gpu_shipped.py tests/test_gpu_geometry.py tests/test_gpu_generate.py tests/test_gpu_timed_mode.py tests/test_gpu_fullsize.py -m gpu -q) > $O/pytest_prefill.log 2>&1; tail -4 $O/pytest_prefill.log
Q="--no-cpu-baseline --no-eager-baseline --skip-extra --no-roofline"
for i in 1 2; do for k in 0 1; do
  VVHIP_G3_KSPLIT=$k timeout 200 python bench.py --workload 1p5b --steps 30 --warmup 5 $Q > $O/ab_g3_1p5b_${k}_$i.json 2>/dev/null
  VVHIP_G3_KSPLIT=$k timeout 200 python bench.py --model 7b --workload 1p5b --solver-steps 10 --steps 30 --warmup 5 $Q > $O/ab_g3_7bshort_${k}_$i.json 2>/dev/null
done; done
for f in $O/ab_g3_*.json; do echo $(basename $f) $(python -c "
import json;d=json.load(open('$f'));print(d['ms_per_step'], d['extra']['prefill_phases'], d['extra']['first_audio']['p50_ms'])"); done
