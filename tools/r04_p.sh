#!/bin/bash
# block1d: several time tiles per workgroup on long inputs (voice-prompt encode): parity tests, then A/B (VVHIP_BLOCK1D_TPW=1 = one tile, as before)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04p; mkdir -p $O
cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_shipped.py tests/test_gpu_fullsize.py -m gpu -q -x -k "encode or codec or voice or acoustic or semantic or block" 2>&1 | tail -4
Q="--no-cpu-baseline --no-eager-baseline --skip-extra --no-roofline --no-parity"
for i in 1 2; do
  for f in 1 8; do
    VVHIP_BLOCK1D_TPW=$f timeout 200 python bench.py --workload 1p5b --steps 10 --warmup 3 $Q > $O/tpw_1p5b_${f}_$i.json 2>/dev/null
  done
done
for f in 1 8 16; do VVHIP_BLOCK1D_TPW=$f timeout 200 python bench.py --steps 5 --warmup 2 $Q > $O/tpw_7b_${f}.json 2>/dev/null; done
for f in $O/*.json; do echo $(basename $f) $(python -c "
import json;d=json.load(open('$f'));print(d['ms_per_step'], d['extra']['prefill_phases']['voice_encode_s'], d['extra']['prefill_phases']['lm_passes_s'], d['extra']['first_audio']['p50_ms'])"); done
