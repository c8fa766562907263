#!/bin/bash
# round 4, pass e: parity of the folded norm+dwconv prologue (codec tests), then its A/B on the three single-utterance configs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04e; mkdir -p $O
cd $R; export TMPDIR=/tmp
Q="--no-cpu-baseline --no-eager-baseline --skip-extra --no-roofline"
(time timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_streaming.py tests/test_gpu_timed_mode.py tests/test_gpu_fulldepth.py -m gpu -q) > $O/pytest_codec.log 2>&1; tail -5 $O/pytest_codec.log
for f in 0 1; do
  VVHIP_FOLD_NORMDW=$f timeout 200 python bench.py --workload 1p5b --steps 150 --warmup 10 $Q > $O/ab_fold_1p5b_$f.json 2>/dev/null
  VVHIP_FOLD_NORMDW=$f timeout 200 python bench.py --workload streaming --steps 60 $Q > $O/ab_fold_streaming_$f.json 2>/dev/null
  VVHIP_FOLD_NORMDW=$f timeout 300 python bench.py --steps 30 --warmup 5 $Q > $O/ab_fold_7b_$f.json 2>/dev/null
done
for f in $O/ab_*.json; do echo $(basename $f) $(python -c "
import json;d=json.load(open('$f'));print(d['value'],d['ms_per_step'])" 2>/dev/null); done
