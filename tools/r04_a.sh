#!/bin/bash
# round 4, pass a: full-depth parity test, the default bench line (now with `parity`), kernel trace + gap report of the 1.5B and 7B decode
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; mkdir -p $O
cd $R; export TMPDIR=/tmp
Q="--no-cpu-baseline --no-eager-baseline --no-roofline --skip-extra"
timeout 600 python -m pytest tests/test_gpu_fulldepth.py -x -q -s -m gpu > $O/fulldepth.log 2>&1; grep -E "full depth|HIP vs|reference bf16|passed|failed|Error" $O/fulldepth.log | head -20
timeout 500 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print("VALUE", d["value"], d["ms_per_step"], d["extra"]["prefill_phases"])
print(json.dumps(d["parity"], indent=1)[:3000])
print(json.dumps(d["extra"]["configs"]["configs[1]"].get("parity"), indent=1)[:3000])
PY
timeout 300 rocprofv3 --kernel-trace -d $O/p15 -o t -- python bench.py --workload 1p5b --steps 60 --warmup 10 $Q > $O/1p5b_under_rocprof.json 2> $O/rp15.err
python tools/rocprof_summary.py $O/p15/t_results.db $O/r04a_1p5b > $O/r04a_1p5b_top.txt 2>&1; rm -rf $O/p15
timeout 300 rocprofv3 --kernel-trace -d $O/p7 -o t -- python bench.py --steps 20 --warmup 5 $Q > $O/7b_under_rocprof.json 2> $O/rp7.err
python tools/rocprof_summary.py $O/p7/t_results.db $O/r04a_7b > $O/r04a_7b_top.txt 2>&1; rm -rf $O/p7
cat $O/r04a_1p5b_gaps.txt | head -30; cat $O/r04a_7b_gaps.txt | head -20
