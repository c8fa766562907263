R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/extra; mkdir -p $O; cd $R; export TMPDIR=/tmp
TAG=r03
timeout 300 rocprofv3 --kernel-trace --stats -d $O/p_15 -o k -- python bench.py --workload 1p5b --no-cpu-baseline --no-eager-baseline --steps 100 --warmup 4 > $O/${TAG}_1p5b_under_rocprof.json 2> $O/rocprof_15.err
python tools/rocprof_summary.py $O/p_15/k_results.db $O/${TAG}_1p5b > $O/${TAG}_1p5b_top.txt 2>&1; rm -rf $O/p_15
B7="--batch 8 --speakers 4 --text-tokens 10569 --steps 20 --warmup 5 --skip-extra --no-cpu-baseline --no-eager-baseline"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/p_b8 -o k -- python bench.py $B7 --no-roofline > /dev/null 2> $O/rocprof_b8.err
python tools/rocprof_summary.py $O/p_b8/k_results.db $O/${TAG}_7b_batch8 > $O/${TAG}_7b_batch8_top.txt 2>&1; rm -rf $O/p_b8
ls $O; head -8 $O/${TAG}_7b_batch8_top.txt | cut -c1-150
