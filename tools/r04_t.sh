#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04t; mkdir -p $O
cd $R; export TMPDIR=/tmp
Q="--no-cpu-baseline --no-eager-baseline --skip-extra --no-roofline --no-parity"
timeout 200 python bench.py --workload 1p5b --steps 3 --warmup 1 $Q > $O/b2.json 2>/dev/null; python -c "
import json;d=json.load(open('$O/b2.json'));print(d['extra']['first_audio']['trials_ms'])"
timeout 200 python -m pytest tests/test_gpu_generate.py tests/test_gpu_kernels.py -m gpu -q -x -k "streamer or pcm or pooled" 2>&1 | grep -E "passed|failed"
