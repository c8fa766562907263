#!/bin/bash
# round 4, pass b: full-depth parity (final verdict form), host-slack sweep, decode-only gap reports, ONE WHOLE UTTERANCE
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04b; mkdir -p $O
cd $R; export TMPDIR=/tmp
Q="--no-cpu-baseline --no-eager-baseline --no-roofline --skip-extra"
timeout 600 python -m pytest tests/test_gpu_fulldepth.py -x -q -s -m gpu > $O/fulldepth.log 2>&1; grep -E "full depth|HIP vs|reference bf16|bounds asserted|passed|failed|Error" $O/fulldepth.log | head -20
for dly in 0 100 200 400 800; do
  timeout 200 python bench.py --workload 1p5b --steps 100 --warmup 10 $Q --host-delay-us $dly > $O/slack_1p5b_$dly.json 2>/dev/null
  timeout 200 python bench.py --steps 30 --warmup 5 $Q --host-delay-us $dly > $O/slack_7b_$dly.json 2>/dev/null
done
for f in $O/slack_*.json; do echo $(basename $f) $(python -c "import json;d=json.load(open('$f'));print(d['ms_per_step'])"); done
timeout 300 rocprofv3 --kernel-trace -d $O/p7 -o t -- python bench.py --steps 20 --warmup 5 $Q > $O/7b_under_rocprof.json 2> $O/rp7.err
python tools/rocprof_summary.py $O/p7/t_results.db $O/r04_7b_northstar > $O/r04_7b_northstar_top.txt 2>&1; rm -rf $O/p7
timeout 300 rocprofv3 --kernel-trace -d $O/p15 -o t -- python bench.py --workload 1p5b --steps 60 --warmup 10 $Q > $O/1p5b_under_rocprof.json 2> $O/rp15.err
python tools/rocprof_summary.py $O/p15/t_results.db $O/r04_1p5b > $O/r04_1p5b_top.txt 2>&1; rm -rf $O/p15
head -3 $O/r04_7b_northstar_timeline.txt; head -12 $O/r04_7b_northstar_gaps.txt; head -3 $O/r04_1p5b_timeline.txt; head -12 $O/r04_1p5b_gaps.txt
timeout 600 python bench.py --full-utterance > $O/r04_full_utterance.json 2> $O/full.err; tail -c 300 $O/full.err
python -c "
import json;d=json.load(open('$O/r04_full_utterance.json'));print(d['value'],d['ms_per_step'],d['steps']);print(json.dumps(d['extra'],indent=1))"
