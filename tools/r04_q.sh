#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04q; mkdir -p $O
cd $R; export TMPDIR=/tmp
Q="--no-cpu-baseline --no-eager-baseline --skip-extra --no-roofline --no-parity"
timeout 300 python tools/ttfa_profile.py --workload 1p5b --steps 5 --warmup 2 $Q > $O/b2.json 2> $O/ttfa_1p5b_2.txt
grep -n "ttfa\]" $O/ttfa_1p5b_2.txt
python -c "
import json;d=json.load(open('$O/b2.json'));print(d['extra']['first_audio'], d['extra']['prefill_phases'])"
timeout 300 python bench.py --steps 5 --warmup 2 $Q > $O/ns.json 2>/dev/null; python -c "
import json;d=json.load(open('$O/ns.json'));print(d['extra']['first_audio'], d['extra']['prefill_phases'])"
timeout 600 python -m pytest tests/test_gpu_generate.py tests/test_gpu_shipped.py -m gpu -q -x -k "generate or voice or continuous or pretrained" 2>&1 | tail -3
