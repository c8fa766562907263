#!/bin/bash
# session G: prefill attention, 8 waves x 2 row tiles vs 4 waves x 4 row tiles
mkdir -p gpurun_out
O=gpurun_out
export TMPDIR=/tmp
(time timeout 600 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_kernels.py -m gpu -q --tb=short 2>&1 | tail -30) > $O/r02g_pytest.log 2>&1
NS="--skip-extra --no-cpu-baseline --no-roofline --steps 5 --warmup 2"
timeout 300 python bench.py $NS > $O/r02g_ns_rt2.json 2>/dev/null
VVHIP_ATTN2_RT4=1 timeout 300 python bench.py $NS > $O/r02g_ns_rt4.json 2>/dev/null
timeout 300 python bench.py $NS > $O/r02g_ns_rt2_b.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r02g_prof -o ns -- python bench.py $NS > $O/r02g_rocprof.json 2> $O/r02g_rocprof.err
python tools/rocprof_summary.py $O/r02g_prof/ns_results.db $O/r02g_7b_prefill > $O/r02g_7b_prefill_top.txt 2>&1
rm -rf $O/r02g_prof
for f in $O/r02g_ns_*.json; do echo $(basename $f) $(python -c "import json;d=json.load(open('$f'));print(d['ms_per_step'], d['extra']['prefill_phases'])"); done
grep -E "passed|failed" $O/r02g_pytest.log
grep -E "attn_prefill2|gemm3|pack_rows|rope_append" $O/r02g_7b_prefill_top.txt | cut -c1-160
