#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
NS="--skip-extra --no-cpu-baseline --no-roofline --steps 5 --warmup 2"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p_ns -o ns -- python bench.py $NS > $O/r02t_ns.json 2> $O/r02t_err.txt
python tools/rocprof_summary.py $O/p_ns/ns_results.db $O/r02t_ns > $O/r02t_ns_top.txt 2>&1; rm -rf $O/p_ns
grep "gemm4\|gemm3\|attn_prefill2\|pack_rows\|rope_append" $O/r02t_ns_kernel_shapes.csv | awk -F, '{printf "%-44s grid=%-8s calls=%-4s avg=%.1f us\n", substr($1,1,44), $(NF-6), $(NF-2), $NF/1000}'
python -c "import json;d=json.load(open('$O/r02t_ns.json'));print(d['extra']['prefill_phases'])"
cat $O/r02t_ns_timeline.txt
