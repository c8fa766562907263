#!/bin/bash
# Round-end measurement suite (run on the GPU box through gpurun): bench JSONs, rocprofv3 kernel stats, PMC traffic,
# step timelines (needs build/variants/libvvhip_t.so from tools/variant.sh t "-DVV_GEMM_TIMING" all).  Outputs: gpurun_out/final/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
timeout 100 python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 2 > /dev/null 2>&1      # page the image in
VVHIP_TIME_PREFILL=1 timeout 400 python bench.py > $O/r01_bench_1p5b.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
VVHIP_TIME_PREFILL=1 timeout 200 python bench.py --model 7b --no-cpu-baseline > $O/r01_7b_short_n10.json 2>/dev/null
timeout 200 python bench.py --model 7b --kv-start 32000 --solver-steps 20 --steps 60 --no-cpu-baseline > $O/r01_7b_32k_n20.json 2>/dev/null
timeout 200 python bench.py --model 7b --kv-start 32000 --steps 60 --no-cpu-baseline > $O/r01_7b_32k_n10.json 2>/dev/null
timeout 200 python bench.py --model 1.5b --kv-start 64000 --solver-steps 20 --steps 60 --no-cpu-baseline > $O/r01_1p5b_64k_n20.json 2>/dev/null
timeout 200 python bench.py --model 0.5b-streaming --no-cpu-baseline > $O/r01_streaming.json 2>/dev/null
timeout 200 python bench.py --model 7b --batch 8 --speakers 4 --no-cpu-baseline --no-roofline --steps 60 > $O/r01_7b_4spk_batch8.json 2>/dev/null
timeout 200 python bench.py --batch 8 --no-cpu-baseline --no-roofline --steps 100 > $O/r01_1p5b_batch8.json 2>/dev/null
VVHIP_TIME_PREFILL=1 timeout 200 python bench.py --model 7b --prefill-rows 1024 --text-tokens 10000 --no-cpu-baseline --no-roofline --steps 20 > $O/r01_7b_prefill_10k.json 2>/dev/null
for m in 1.5b 7b; do
  VVHIP_TIMELINE=$O/tl_$m.npz VVHIP_LIB=build/variants/libvvhip_t.so timeout 200 python bench.py --model $m --no-cpu-baseline --no-roofline --steps 20 --warmup 4 > /dev/null 2>&1
  python tools/step_timeline.py $O/tl_$m.npz > $O/r01_${m}_step_timeline.txt 2>/dev/null; rm -f $O/tl_$m.npz
done
cd /tmp && export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o ks -- python $R/bench.py --no-cpu-baseline --no-roofline --steps 100 --warmup 4 > /tmp/prof.log 2>&1
cp /tmp/prof/ks_kernel_stats.csv $O/r01_1p5b_bench_kernel_stats.csv
timeout 250 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc -o pm -- python $R/bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 4 > /tmp/pmc.log 2>&1
python $R/tools/pmc_traffic.py /tmp/pmc/pm_counter_collection.csv 1.5b 26 $O/r01_1p5b_pmc_fetch_size_by_kernel.csv > /dev/null
cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
for f in $O/r01_*.json; do echo $(basename $f) $(python -c "import json,sys; d=json.load(open('$f')); e=d.get('extra') or {}; print(d.get('value'), d.get('ms_per_step'), (d.get('roofline') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('value'), e.get('prefill_phases'), e.get('p50_first_audio_ms'))" 2>/dev/null); done
python - <<PY
import csv, json
rows = list(csv.DictReader(open("$O/r01_1p5b_bench_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows if "vv_gemv_kernel" in r["Name"]); calls = sum(int(r["Calls"]) for r in rows if "vv_gemv_kernel" in r["Name"])
allk = sum(float(r["TotalDurationNs"]) for r in rows)
json.dump({"1.5b": {"kernel": "vv_gemv_kernel (all instantiations)", "avg_launch_us": round(tot / calls / 1e3, 3), "dispatches": calls,
                    "sum_all_kernels_ms": round(allk / 1e6, 2), "note": "rocprofv3 --kernel-trace --stats of the default bench; under hipGraph replay per-kernel intervals overlap (their sum exceeds the wall time), so this is an upper bound on a launch's own duration"}},
          open("$O/rocprof_gemv.json", "w"), indent=1)
PY
