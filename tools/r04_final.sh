#!/bin/bash
# The round's closing measurement pass (a trimmed tools/r04_measure.sh: what changes with the build id).  Bench lines of every
# single-GPU config, FETCH_SIZE passes -> profiles/pmc_traffic.json at this build, the default line again (now with `traffic`),
# kernel stats + timeline of the north-star run, MFMA-busy pass of the prefill kernels, then the GPU test log.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R; export TMPDIR=/tmp
TAG=${1:-r04}
Q="--no-cpu-baseline --no-eager-baseline"
timeout 300 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_default.json 2> $O/bench_default.err; tail -c 200 $O/bench_default.err
timeout 200 python bench.py --workload 1p5b --steps 150 --warmup 10 $Q > $O/${TAG}_1p5b.json 2>/dev/null
timeout 200 python bench.py --workload streaming --steps 60 > $O/${TAG}_streaming.json 2>/dev/null
bash tools/pmc_refresh.sh $TAG $O > $O/pmc_refresh.log 2>&1; tail -3 $O/pmc_refresh.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_default.json 2> $O/bench_default.err
B7="--batch 8 --speakers 4 --text-tokens 10569 --steps 20 --warmup 5 --skip-extra $Q"
timeout 400 python bench.py $B7 > $O/${TAG}_7b_4spk_batch8_32k.json 2>/dev/null
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --skip-extra $Q > $O/${TAG}_torchrun_n1.json 2> $O/torchrun_n1.err
timeout 200 python bench.py --model 7b --workload 1p5b --solver-steps 10 --steps 60 $Q > $O/${TAG}_7b_short_n10.json 2>/dev/null
timeout 200 python bench.py --workload 1p5b --kv-start 64000 --solver-steps 20 --steps 60 $Q > $O/${TAG}_1p5b_64k_n20.json 2>/dev/null
timeout 200 python bench.py --workload 1p5b --batch 8 --steps 100 $Q > $O/${TAG}_1p5b_batch8.json 2>/dev/null
timeout 200 python bench.py --workload 1p5b --batch 4 --continuous 12 --steps 40 --warmup 5 $Q --no-roofline > $O/${TAG}_1p5b_continuous.json 2>/dev/null
NS="--skip-extra $Q --steps 20 --warmup 5"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/p_ns -o ns -- python bench.py $NS > $O/${TAG}_bench_under_rocprof.json 2> $O/rocprof_ns.err
python tools/rocprof_summary.py $O/p_ns/ns_results.db $O/${TAG}_7b_northstar --around vv_attn_prefill4 40 > $O/${TAG}_7b_northstar_top.txt 2>&1; rm -rf $O/p_ns
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $O/p_mfma -o m -- python bench.py --skip-extra $Q --no-roofline --steps 2 --warmup 1 > /dev/null 2> $O/pmc_mfma.err
python tools/rocprof_summary.py $O/p_mfma/m_results.db $O/${TAG}_7b_pmc_mfma --pmc > $O/${TAG}_7b_pmc_mfma_top.txt 2>&1; rm -rf $O/p_mfma
timeout 600 python bench.py --full-utterance > $O/${TAG}_full_utterance.json 2> $O/full.err
(time timeout 700 python -m pytest tests -m gpu -q) > $O/${TAG}_pytest_gpu.log 2>&1; tail -3 $O/${TAG}_pytest_gpu.log
for f in $O/${TAG}_*.json; do echo $(basename $f) $(python -c "
import json,sys
t=open('$f').read(); l=[x for x in t.split('\n') if x.startswith('{')]
d=json.loads(l[-1]); e=d.get('extra') or {}; r=d.get('roofline') or {}
print(d.get('value'), d.get('ms_per_step'), r.get('frac'), r.get('traffic'), (r.get('attention') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('value'), e.get('prefill_phases'), (e.get('first_audio') or {}).get('p50_ms'), e.get('p50_first_audio_ms'))" 2>/dev/null); done
du -sh $O
