#!/bin/bash
# round 4, pass c: the GPU suite on the new attention combine / 8-wave form and the slot-batched tile GEMM, then their A/Bs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O
cd $R; export TMPDIR=/tmp
Q="--no-cpu-baseline --no-eager-baseline --skip-extra"
(time timeout 900 python -m pytest tests -m gpu -q -x) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
for w8 in 0 160; do
  VVHIP_ATTN_W8_MIN=$w8 timeout 200 python bench.py --workload 1p5b --steps 150 --warmup 10 $Q > $O/ab_attn_1p5b_w8min$w8.json 2>/dev/null
  VVHIP_ATTN_W8_MIN=$w8 timeout 200 python bench.py --workload streaming --steps 60 $Q > $O/ab_attn_streaming_w8min$w8.json 2>/dev/null
done
B7="--batch 8 --speakers 4 --text-tokens 10569 --steps 20 --warmup 5 $Q"
for mw in 1000000 40 16; do
  VVHIP_TILE_MIN_WGS_SLOTS=$mw timeout 400 python bench.py $B7 > $O/ab_tile_7b_batch8_minwgs$mw.json 2>/dev/null
done
for mw in 1000000 40; do
  VVHIP_TILE_MIN_WGS_SLOTS=$mw timeout 200 python bench.py --workload 1p5b --batch 8 --steps 100 $Q > $O/ab_tile_1p5b_batch8_minwgs$mw.json 2>/dev/null
done
for f in $O/ab_*.json; do echo $(basename $f) $(python -c "
import json;d=json.load(open('$f'));r=d.get('roofline') or {};a=r.get('attention') or {}
print(d['value'],d['ms_per_step'],'attn_us',a.get('avg_launch_us'),'gemv_us',r.get('avg_launch_us'))" 2>/dev/null); done
