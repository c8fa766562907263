#!/bin/bash
# round 4, pass d: the whole GPU suite on the current build, then a kernel trace of batch-8 decode (what is left in the tokenizer chains)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04d; mkdir -p $O
cd $R; export TMPDIR=/tmp
Q="--no-cpu-baseline --no-eager-baseline --skip-extra"
(time timeout 900 python -m pytest tests -m gpu -q) > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
B7="--batch 8 --speakers 4 --text-tokens 10569 --steps 20 --warmup 5 $Q --no-roofline"
timeout 400 rocprofv3 --kernel-trace -d $O/p8 -o t -- python bench.py $B7 > $O/7b_batch8_under_rocprof.json 2> $O/rp8.err
python tools/rocprof_summary.py $O/p8/t_results.db $O/r04_7b_batch8 > $O/r04_7b_batch8_top.txt 2>&1; rm -rf $O/p8
cat $O/r04_7b_batch8_timeline.txt
