#!/bin/bash
# FETCH_SIZE pass of the binary at HEAD for the three single-GPU configs -> profiles/r0N_*_pmc_fetch_pmc_by_kernel.csv and
# profiles/pmc_traffic.json (tools/pmc_refresh.py).  Run on the GPU box through gpurun; counters in their own pass (no --stats,
# no other trace domain).  $1 = tag (r03), $2 = directory with the bench JSON lines of the same build (optional).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc; mkdir -p $O; cd $R; export TMPDIR=/tmp
TAG=${1:-r03}; B=${2:-$R/gpurun_out/final}
Q="--skip-extra --no-cpu-baseline --no-eager-baseline --no-roofline"
run() {   # key, bench args, bench json
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/p_$1 -o f -- python bench.py $Q $2 > /dev/null 2> $O/pmc_$1.err
  python tools/rocprof_summary.py $O/p_$1/f_results.db $O/${TAG}_$1_pmc_fetch --pmc > $O/${TAG}_$1_pmc_fetch_top.txt 2>&1; rm -rf $O/p_$1
  python tools/pmc_refresh.py $3 $O/${TAG}_$1_pmc_fetch_pmc_by_kernel.csv $4 | tail -25
}
run 7b "--steps 40 --warmup 2" 7b $B/${TAG}_bench_default.json
run 1p5b "--workload 1p5b --steps 100 --warmup 2" 1.5b $B/${TAG}_1p5b.json
run streaming "--workload streaming --steps 60" 0.5b-streaming $B/${TAG}_streaming.json
# the configs[3] per-GPU unit (8 utterances in lock-step: vv_gemv16p_kernel + batch attention) under its own key
run 7b_batch8 "--batch 8 --speakers 4 --text-tokens 10569 --steps 20 --warmup 2" 7b-batch8 $B/${TAG}_7b_4spk_batch8_32k.json
cp profiles/pmc_traffic.json $O/pmc_traffic.json
