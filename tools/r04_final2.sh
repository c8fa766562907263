#!/bin/bash
# round 4 closing pass on the FINAL build: bench lines, FETCH_SIZE passes -> profiles/pmc_traffic.json,
# the default line again (now with `traffic`), north-star kernel trace + MFMA-busy pass, the GPU suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final3; mkdir -p $O
cd $R; export TMPDIR=/tmp
TAG=r04
Q="--no-cpu-baseline --no-eager-baseline"
timeout 300 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_default.json 2> $O/bench_default.err; tail -c 200 $O/bench_default.err
timeout 200 python bench.py --workload 1p5b --steps 150 --warmup 10 $Q > $O/${TAG}_1p5b.json 2>/dev/null
timeout 200 python bench.py --workload streaming --steps 60 > $O/${TAG}_streaming.json 2>/dev/null
bash tools/pmc_refresh.sh $TAG $O > $O/pmc_refresh.log 2>&1; tail -3 $O/pmc_refresh.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_default.json 2> $O/bench_default.err
NS="--skip-extra $Q --steps 20 --warmup 5"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/p_ns -o ns -- python bench.py $NS > $O/${TAG}_bench_under_rocprof.json 2> $O/rocprof_ns.err
python tools/rocprof_summary.py $O/p_ns/ns_results.db $O/${TAG}_7b_northstar --around vv_attn_prefill4 40 > $O/${TAG}_7b_northstar_top.txt 2>&1; rm -rf $O/p_ns
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $O/p_mfma -o m -- python bench.py --skip-extra $Q --no-roofline --steps 2 --warmup 1 > /dev/null 2> $O/pmc_mfma.err
python tools/rocprof_summary.py $O/p_mfma/m_results.db $O/${TAG}_7b_pmc_mfma --pmc > $O/${TAG}_7b_pmc_mfma_top.txt 2>&1; rm -rf $O/p_mfma
(time timeout 700 python -m pytest tests -m gpu -q) > $O/${TAG}_pytest_gpu.log 2>&1; tail -3 $O/${TAG}_pytest_gpu.log
python - <<PY
import json
d=json.load(open("$O/${TAG}_bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"].get("whole_step_achieved_frac"), d["extra"]["prefill_phases"], d["extra"]["first_audio"]["p50_ms"])
print(json.dumps({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ("latent","pos_hidden","neg_hidden","frame_rms_db","within_bounds","within_survey_bounds")}) for k,v in d["parity"].items() if k not in ("definition",)},indent=0))
c=d["extra"]["configs"]
print(c["configs[1]"]["ms_per_step"], c["configs[4]"]["ms_per_step"], (c["configs[1]"].get("parity") or {}).get("within_bounds"))
PY
