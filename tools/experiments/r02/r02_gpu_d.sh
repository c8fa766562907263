#!/bin/bash
# session D: decode-attention layout sweep (cyclic vs contiguous block assignment) + per-shape kernel stats
mkdir -p gpurun_out
O=gpurun_out
export TMPDIR=/tmp
(time timeout 600 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_kernels.py tests/test_gpu_generate.py -m gpu -q --tb=short 2>&1 | tail -40) > $O/r02d_pytest.log 2>&1
NS="--skip-extra --no-cpu-baseline --no-roofline --steps 20 --warmup 5"
for sp in 512 1024 2048; do
  VVHIP_ATTN_SPLIT_POS=$sp VVHIP_ATTN_LONG=100000000 timeout 300 python bench.py $NS > $O/r02d_ns_cyc_$sp.json 2>/dev/null
  VVHIP_ATTN_CONTIGUOUS=1 VVHIP_ATTN_SPLIT_POS=$sp VVHIP_ATTN_LONG=100000000 timeout 300 python bench.py $NS > $O/r02d_ns_con_$sp.json 2>/dev/null
done
VVHIP_ATTN_SPLIT_POS=256 VVHIP_ATTN_LONG=100000000 timeout 300 python bench.py $NS > $O/r02d_ns_cyc_256.json 2>/dev/null
VVHIP_ATTN_SPLIT_POS=512 VVHIP_ATTN_LONG=4096 timeout 300 python bench.py $NS > $O/r02d_ns_cyc_512_w8.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r02d_prof -o ns -- python bench.py --skip-extra --no-cpu-baseline --steps 20 --warmup 5 > $O/r02d_bench_rocprof.json 2> $O/r02d_rocprof.err
python tools/rocprof_summary.py $O/r02d_prof/ns_results.db $O/r02d_7b_northstar > $O/r02d_7b_northstar_top.txt 2>&1
rm -rf $O/r02d_prof
for f in $O/r02d_ns_*.json; do echo $(basename $f) $(python -c "import json;d=json.load(open('$f'));print(d['ms_per_step'])"); done
tail -3 $O/r02d_pytest.log
