#!/bin/bash
# fused reduce + RMSNorm + pack tail of the short-prompt projections, pack_rows with four k-tiles per trip: parity, then A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04o; mkdir -p $O
cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_shipped.py tests/test_gpu_geometry.py tests/test_gpu_kernels.py -m gpu -q -x -k "prefill or pack or short_prompt or gemm3" 2>&1 | tail -4
Q="--no-cpu-baseline --no-eager-baseline --skip-extra --no-roofline --no-parity"
for i in 1 2; do
  for f in 0 1; do
    VVHIP_G3_FUSE_PACK=$f timeout 200 python bench.py --workload 1p5b --steps 10 --warmup 3 $Q > $O/fuse_1p5b_${f}_$i.json 2>/dev/null
    VVHIP_G3_FUSE_PACK=$f timeout 200 python bench.py --model 7b --workload 1p5b --solver-steps 10 --steps 10 --warmup 3 $Q > $O/fuse_7bshort_${f}_$i.json 2>/dev/null
  done
  timeout 200 python bench.py --steps 5 --warmup 2 $Q > $O/ns_new_$i.json 2>/dev/null
done
for f in $O/*.json; do echo $(basename $f) $(python -c "
import json;d=json.load(open('$f'));print(d['extra']['libvvhip_build_id'], d['ms_per_step'], d['extra']['prefill_phases']['lm_passes_s'], d['extra']['first_audio']['p50_ms'])"); done
