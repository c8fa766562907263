R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
timeout 400 python bench.py > $O/r01_bench_1p5b.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.err
timeout 200 python bench.py --model 7b --no-cpu-baseline > $O/r01_7b_short_n10.json 2>/dev/null
timeout 200 python bench.py --model 7b --kv-start 32000 --solver-steps 20 --steps 60 --no-cpu-baseline > $O/r01_7b_32k_n20.json 2>/dev/null
timeout 200 python bench.py --model 7b --kv-start 32000 --steps 60 --no-cpu-baseline > $O/r01_7b_32k_n10.json 2>/dev/null
timeout 200 python bench.py --model 1.5b --kv-start 64000 --solver-steps 20 --steps 60 --no-cpu-baseline > $O/r01_1p5b_64k_n20.json 2>/dev/null
timeout 200 python bench.py --model 0.5b-streaming --no-cpu-baseline > $O/r01_streaming.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o ks -- python $R/bench.py --no-cpu-baseline --no-roofline --steps 100 --warmup 4 > /tmp/prof.log 2>&1
cp /tmp/prof/ks_kernel_stats.csv $O/r01_1p5b_bench_kernel_stats.csv
timeout 250 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc -o pm -- python $R/bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 4 > /tmp/pmc.log 2>&1
ls /tmp/pmc
python $R/tools/pmc_traffic.py /tmp/pmc/pm_counter_collection.csv 1.5b 26 $O/r01_1p5b_pmc_fetch_size_by_kernel.csv
cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
for f in $O/*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d.get('value'), d.get('ms_per_step'), (d.get('roofline') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('value'))" 2>/dev/null; done
