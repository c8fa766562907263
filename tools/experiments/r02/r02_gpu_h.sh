#!/bin/bash
# session H: GEMM3 workgroup height (128 vs 256 rows) A/B
mkdir -p gpurun_out
O=gpurun_out
export TMPDIR=/tmp
(time timeout 600 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_kernels.py -m gpu -q --tb=short 2>&1 | tail -30) > $O/r02h_pytest.log 2>&1
(VVHIP_GEMM3_TR=8 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "gemm3" 2>&1 | tail -8) > $O/r02h_pytest_tr8.log 2>&1
NS="--skip-extra --no-cpu-baseline --no-roofline --steps 5 --warmup 2"
for tr in 4 8 0; do
  VVHIP_GEMM3_TR=$tr timeout 300 python bench.py $NS > $O/r02h_ns_tr$tr.json 2>/dev/null
done
VVHIP_GEMM3_TR=8 timeout 600 rocprofv3 --kernel-trace --stats -d $O/r02h_prof -o ns -- python bench.py $NS > $O/r02h_rocprof.json 2> $O/r02h_rocprof.err
python tools/rocprof_summary.py $O/r02h_prof/ns_results.db $O/r02h_7b_prefill_tr8 > $O/r02h_top.txt 2>&1
rm -rf $O/r02h_prof
for f in $O/r02h_ns_*.json; do echo $(basename $f) $(python -c "import json;d=json.load(open('$f'));print(d['ms_per_step'], d['extra']['prefill_phases'])"); done
grep -E "passed|failed" $O/r02h_pytest.log $O/r02h_pytest_tr8.log
grep -E "attn_prefill2|gemm3|pack_rows" $O/r02h_top.txt | cut -c1-160
