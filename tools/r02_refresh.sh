#!/bin/bash
# Round-2 late refresh at HEAD (gemm4 / batch decode / batched codec chain): GPU suite, default bench line, batch-8 line,
# rocprof kernel stats of both, PMC MFMA pass.  Ordered by importance: the call may be cut at its time limit.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R; export TMPDIR=/tmp
TAG=r02
(time timeout 420 python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
(time timeout 400 python bench.py --steps 20 --warmup 5) > $O/${TAG}_bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
B7="--batch 8 --speakers 4 --text-tokens 10569 --steps 20 --warmup 5 --skip-extra --no-cpu-baseline"
timeout 400 python bench.py $B7 > $O/${TAG}_7b_4spk_batch8_32k.json 2>/dev/null
NS="--skip-extra --no-cpu-baseline --steps 20 --warmup 5"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/p_ns -o ns -- python bench.py $NS > $O/${TAG}_bench_under_rocprof.json 2> $O/rocprof_ns.err
python tools/rocprof_summary.py $O/p_ns/ns_results.db $O/${TAG}_7b_northstar --around vv_attn_prefill3 40 > $O/${TAG}_7b_northstar_top.txt 2>&1; rm -rf $O/p_ns
timeout 400 rocprofv3 --kernel-trace --stats -d $O/p_b8 -o k -- python bench.py $B7 --no-roofline > /dev/null 2> $O/rocprof_b8.err
python tools/rocprof_summary.py $O/p_b8/k_results.db $O/${TAG}_7b_batch8 > $O/${TAG}_7b_batch8_top.txt 2>&1; rm -rf $O/p_b8
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $O/p_mfma -o m -- python bench.py --skip-extra --no-cpu-baseline --no-roofline --steps 2 --warmup 1 > /dev/null 2> $O/pmc_mfma.err
python tools/rocprof_summary.py $O/p_mfma/m_results.db $O/${TAG}_7b_pmc_mfma --pmc > $O/${TAG}_7b_pmc_mfma_top.txt 2>&1; rm -rf $O/p_mfma
timeout 200 python bench.py --workload 1p5b --steps 150 --warmup 10 --no-cpu-baseline > $O/${TAG}_1p5b.json 2>/dev/null
timeout 200 python bench.py --workload 1p5b --batch 8 --steps 100 --no-cpu-baseline --no-roofline > $O/${TAG}_1p5b_batch8.json 2>/dev/null
Q="--skip-extra --no-cpu-baseline --no-roofline --steps 20 --warmup 5"
for sp in 512 768; do VVHIP_ATTN_SPLIT_POS=$sp timeout 200 python bench.py $Q > $O/ab_split$sp.json 2>/dev/null; echo split_pos $sp $(python -c "import json; print(json.load(open('$O/ab_split$sp.json'))['ms_per_step'])"); done
timeout 200 python bench.py --workload 1p5b --kv-start 64000 --solver-steps 20 --steps 60 --no-cpu-baseline --no-roofline > $O/${TAG}_1p5b_64k_n20.json 2>/dev/null
VVHIP_ATTN_SPLIT_POS=512 timeout 200 python bench.py --workload 1p5b --kv-start 64000 --solver-steps 20 --steps 60 --no-cpu-baseline --no-roofline > $O/ab_1p5b_64k_split512.json 2>/dev/null
for f in $O/${TAG}_*.json $O/ab_*.json; do echo $(basename $f) $(python -c "import json,sys; d=json.load(open('$f')); e=d.get('extra') or {}; print(d.get('value'), d.get('ms_per_step'), (d.get('roofline') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('value'), e.get('prefill_phases'), e.get('p50_first_audio_ms'))" 2>/dev/null); done
du -sh $O
