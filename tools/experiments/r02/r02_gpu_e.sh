#!/bin/bash
# session E: batched split merge + prefill-attention rescale skip: parity, split sweep, batch-8 profile
mkdir -p gpurun_out
O=gpurun_out
export TMPDIR=/tmp
(time timeout 600 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_kernels.py tests/test_gpu_generate.py tests/test_gpu_timed_mode.py -m gpu -q --tb=short 2>&1 | tail -40) > $O/r02e_pytest.log 2>&1
NS="--skip-extra --no-cpu-baseline --no-roofline --steps 20 --warmup 5"
for sp in 256 512 1024; do
  VVHIP_ATTN_SPLIT_POS=$sp timeout 300 python bench.py $NS > $O/r02e_ns_split_$sp.json 2>/dev/null
done
B8="--batch 8 --speakers 4 --text-tokens 10569 --steps 20 --warmup 5 --skip-extra --no-cpu-baseline --no-roofline"
for sp in 512 1024; do
  VVHIP_ATTN_SPLIT_POS=$sp timeout 600 python bench.py $B8 > $O/r02e_b8_split_$sp.json 2>/dev/null
done
timeout 900 rocprofv3 --kernel-trace --stats -d $O/r02e_prof -o b8 -- python bench.py $B8 > $O/r02e_b8_rocprof.json 2> $O/r02e_rocprof.err
python tools/rocprof_summary.py $O/r02e_prof/b8_results.db $O/r02e_7b_batch8 > $O/r02e_7b_batch8_top.txt 2>&1
rm -rf $O/r02e_prof
for f in $O/r02e_ns_*.json $O/r02e_b8_split*.json; do echo $(basename $f) $(python -c "import json;d=json.load(open('$f'));print(d['ms_per_step'], d['extra'].get('prefill_phases'))"); done
tail -3 $O/r02e_pytest.log
