#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "prefill" 2>&1 | grep -v "^$" | tail -6 > $O/r02r_tests.txt
cat $O/r02r_tests.txt
for mode in default 0 all; do
  if [ $mode = default ]; then unset VVHIP_GEMM4; else export VVHIP_GEMM4=$mode; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/p_g -o k -- python tools/probe_gemm3.py > $O/r02r_probe_$mode.txt 2>$O/r02r_err_$mode.txt
  python tools/rocprof_summary.py $O/p_g/k_results.db $O/r02r_g_$mode > /dev/null 2>&1; rm -rf $O/p_g
  echo "== mode $mode"; grep "vv_gemm" $O/r02r_g_${mode}_kernel_shapes.csv | awk -F, '{printf "%-40s grid=%-8s calls=%-4s avg=%.1f us\n", substr($1,1,40), $(NF-6), $(NF-2), $NF/1000}'
done
unset VVHIP_GEMM4
