#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B15="--workload 1p5b --batch 8 --steps 60 --no-cpu-baseline --no-roofline"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/p_b15 -o k -- python bench.py $B15 > $O/r02n_1p5b_b8.json 2>$O/r02n_err1.txt
python tools/rocprof_summary.py $O/p_b15/k_results.db $O/r02n_1p5b_b8 > $O/r02n_1p5b_b8_top.txt 2>&1; rm -rf $O/p_b15
B7="--batch 8 --speakers 4 --text-tokens 10569 --steps 20 --warmup 5 --skip-extra --no-cpu-baseline --no-roofline"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p_b7 -o k -- python bench.py $B7 > $O/r02n_7b_b8.json 2>$O/r02n_err2.txt
python tools/rocprof_summary.py $O/p_b7/k_results.db $O/r02n_7b_b8 > $O/r02n_7b_b8_top.txt 2>&1; rm -rf $O/p_b7
head -40 $O/r02n_7b_b8_top.txt
du -sh $O
