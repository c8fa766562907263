#!/bin/bash
# round 4, pass f: the loader / consumer experiment (tools/experiments/loader_consumer), then the default line on the rebuilt library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04f; mkdir -p $O
cd $R; export TMPDIR=/tmp
for a in "test 2" "1.5b 10" "7b 20"; do
  timeout 180 python tools/experiments/loader_consumer/bench_lc.py $a > $O/lc_$(echo $a | tr ' ' '_').json 2> $O/lc_$(echo $a | tr ' ' '_').err
  echo "== $a rc=$?"; tail -c 1500 $O/lc_$(echo $a | tr ' ' '_').json; tail -3 $O/lc_$(echo $a | tr ' ' '_').err
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eager-baseline --skip-extra > $O/bench_quick.json 2>/dev/null
python -c "
import json;d=json.load(open('$O/bench_quick.json'));print(d['value'],d['ms_per_step'],d['extra']['prefill_phases'],d['extra']['first_audio'])"
