#!/bin/bash
# The round-end measurement suite on one MI355X box (run through gpurun; ~25 GPU-minutes):   bash tools/round_end.sh <tag> [part ...]
# parts (default: all but utterance): tests bench configs traces pmc dist lanes cpuwin utterance
# Everything lands under gpurun_out/<tag>_final/ with the file names profiles/ uses (<tag>_*); copy what is to be judged into profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r06}; shift; PARTS=${*:-tests bench configs traces pmc dist lanes cpuwin}
O=$R/gpurun_out/${TAG}_final; mkdir -p $O; cd $R; export TMPDIR=/tmp
Q="--no-cpu-baseline --no-eager-baseline --skip-extra --no-parity"
has() { [[ " $PARTS " == *" $1 "* ]]; }
say() { echo "== $*"; }
if has tests; then
  say "pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -s > $O/${TAG}_pytest_gpu.log 2>&1; tail -3 $O/${TAG}_pytest_gpu.log
  timeout 300 python __graft_entry__.py smoke > $O/${TAG}_smoke.log 2>&1; tail -1 $O/${TAG}_smoke.log
fi
if has bench; then
  say "bench.py (the driver's command)"; timeout 900 python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err; head -c 400 $O/${TAG}_bench_default.json; echo
fi
if has configs; then
  say "the other configurations"
  timeout 300 python bench.py --workload 1p5b --steps 60 --warmup 10 $Q > $O/${TAG}_1p5b.json 2> $O/e1.err
  timeout 300 python bench.py --workload streaming --steps 60 > $O/${TAG}_streaming.json 2> $O/e2.err
  timeout 600 python bench.py --batch 8 --speakers 4 --text-tokens 10569 --steps 20 --warmup 5 $Q > $O/${TAG}_7b_4spk_batch8_32k.json 2> $O/e3.err
  timeout 300 python bench.py --workload 1p5b --batch 8 --steps 100 --warmup 10 $Q > $O/${TAG}_1p5b_batch8.json 2> $O/e4.err
  timeout 300 python bench.py --workload 1p5b --solver-steps 20 --kv-start 64000 --steps 60 --warmup 5 $Q > $O/${TAG}_1p5b_64k_n20.json 2> $O/e5.err
  timeout 300 python bench.py --workload 1p5b --model 7b --steps 60 --warmup 5 $Q > $O/${TAG}_7b_short_n10.json 2> $O/e6.err
  timeout 300 python bench.py --workload 1p5b --batch 4 --continuous 12 --steps 40 --warmup 5 $Q --no-roofline > $O/${TAG}_1p5b_continuous.json 2> $O/e7.err
  for f in 1p5b streaming 7b_4spk_batch8_32k 1p5b_batch8 1p5b_64k_n20 7b_short_n10 1p5b_continuous; do python - <<PY
import json
try:
    d=json.load(open("$O/${TAG}_$f.json")); r=d.get("roofline") or {}
    print("$f", d["value"], d["ms_per_step"], "frac", r.get("frac"), "whole", r.get("whole_step_achieved_frac"))
except Exception as e: print("$f FAILED", e)
PY
  done
fi
if has traces; then
  say "rocprofv3 --kernel-trace"
  for spec in "7b_northstar|--steps 20 --warmup 5" "1p5b|--workload 1p5b --steps 60 --warmup 10" "7b_batch8|--batch 8 --speakers 4 --text-tokens 10569 --steps 20 --warmup 5"; do
    name=${spec%%|*}; a=${spec#*|}
    timeout 600 rocprofv3 --kernel-trace -d $O/p_$name -o t -- python bench.py $a $Q --no-roofline > $O/${TAG}_${name}_under_rocprof.json 2> $O/rp_$name.err
    python tools/rocprof_summary.py $O/p_$name/t_results.db $O/${TAG}_$name > $O/${TAG}_${name}_top.txt 2>&1; rm -rf $O/p_$name; head -12 $O/${TAG}_${name}_top.txt
  done
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_stats -o s -- python bench.py --steps 20 --warmup 5 $Q --no-roofline > /dev/null 2> $O/rp_stats.err
  cp $O/p_stats/*/s_kernel_stats.csv $O/${TAG}_bench_default_rocprofv3_stats.csv 2>/dev/null || find $O/p_stats -name "*kernel_stats*" -exec cp {} $O/${TAG}_bench_default_rocprofv3_stats.csv \; ; rm -rf $O/p_stats
fi
if has pmc; then
  say "FETCH_SIZE passes -> profiles/pmc_traffic.json"; bash tools/pmc_refresh.sh $TAG $O | tail -30; cp profiles/pmc_traffic.json $O/pmc_traffic.json
  cp gpurun_out/pmc/${TAG}_*_pmc_fetch_* $O/ 2>/dev/null
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $O/p_mfma -o f -- python bench.py $Q --no-roofline --steps 10 --warmup 2 > /dev/null 2> $O/pmc_mfma.err
  python tools/rocprof_summary.py $O/p_mfma/f_results.db $O/${TAG}_7b_pmc_mfma --pmc > $O/${TAG}_7b_pmc_mfma_top.txt 2>&1; rm -rf $O/p_mfma; head -8 $O/${TAG}_7b_pmc_mfma_top.txt
fi
if has dist; then
  say "torchrun with one rank (RCCL) and two ranks time-sharing the GPU (gloo: control flow only)"
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 $Q > $O/${TAG}_torchrun_n1.json 2> $O/d1.err
  VVHIP_BENCH_SHARED_GPU=1 timeout 600 python bench.py --gpus 2 --workload 1p5b --steps 40 $Q > $O/${TAG}_shared_gpu_n2_controlflow.json 2> $O/d2.err
  python - <<PY
import json
for f in ("torchrun_n1","shared_gpu_n2_controlflow"):
    try:
        d=json.load(open("$O/${TAG}_%s.json"%f)); print(f, d["value"], d["n_gpus"], json.dumps(d["extra"].get("rccl"))[:600])
    except Exception as e: print(f, "FAILED", e)
PY
fi
if has lanes; then
  say "generate_interleaved: 1 / 2 / 4 engine contexts over one weight copy"
  L="--workload 1p5b --batch 8 --continuous 32 --steps 100 --warmup 5 $Q --no-roofline"
  for n in 1 2 4 1 2; do timeout 300 python bench.py $L --lanes $n >> $O/${TAG}_lanes_runs.jsonl 2>> $O/lanes.err; done
  python - <<PY
import json
for ln in open("$O/${TAG}_lanes_runs.jsonl"):
    d=json.loads(ln); c=d["extra"]["continuous"]; print("lanes", c.get("lanes",1), d["value"], d["extra"]["utterance_wall_s"], c.get("capture_fallbacks"))
PY
fi
if has cpuwin; then
  say "CPU baseline windows (SURVEY 8d): 16 frames at three KV lengths"; timeout 1200 python bench.py --cpu-windows --cpu-frames 16 > $O/${TAG}_cpu_baseline.json 2> $O/cpuwin.err; tail -4 $O/cpuwin.err; head -c 600 $O/${TAG}_cpu_baseline.json; echo
fi
if has utterance; then
  say "one whole utterance (configs[2] to the reference's length cap: ~3 minutes)"; timeout 900 python bench.py --full-utterance > $O/${TAG}_full_utterance.json 2> $O/utt.err; head -c 500 $O/${TAG}_full_utterance.json; echo
fi
