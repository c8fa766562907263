#!/bin/bash
# last pass of round 4 (library build unchanged since tools/r04_final2.sh: host-side work only): the default line, smoke, the GPU suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final4; mkdir -p $O
cd $R; export TMPDIR=/tmp
(time timeout 400 python bench.py) > $O/r04_bench_default.json 2> $O/bench_default.err; tail -4 $O/bench_default.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
(time timeout 800 python -m pytest tests -m gpu -q) > $O/r04_pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/r04_pytest_gpu.log | tail -3
python - <<PY
import json
d=json.load(open("$O/r04_bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["extra"]["prefill_phases"], d["extra"]["first_audio"]["trials_ms"], d["parity"]["within_bounds"])
c=d["extra"]["configs"]
print({k:(v.get("ms_per_step"), (v.get("first_audio") or {}).get("trials_ms"), v.get("p50_first_audio_ms")) for k,v in c.items()})
PY
