#!/bin/bash
# first audio after the host-side work: index uploads ahead of the encoder, pooled pinned ring buffers, warm-up through a streamer
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04r; mkdir -p $O
cd $R; export TMPDIR=/tmp
Q="--no-cpu-baseline --no-eager-baseline --skip-extra --no-roofline --no-parity"
timeout 300 python bench.py --workload 1p5b --steps 5 --warmup 2 $Q > $O/b_1p5b.json 2>/dev/null
timeout 300 python bench.py --model 7b --workload 1p5b --solver-steps 10 --steps 5 --warmup 2 $Q > $O/b_7bshort.json 2>/dev/null
timeout 300 python bench.py --steps 5 --warmup 2 $Q > $O/b_ns.json 2>/dev/null
timeout 300 python bench.py --workload streaming --steps 30 $Q > $O/b_streaming.json 2>/dev/null
for f in $O/b_*.json; do echo $(basename $f) $(python -c "
import json;d=json.load(open('$f'));e=d['extra'];print(d['ms_per_step'], e.get('first_audio') and e['first_audio']['trials_ms'], e.get('p50_first_audio_ms'), e.get('prefill_phases'))"); done
(timeout 800 python -m pytest tests/test_gpu_generate.py tests/test_gpu_shipped.py tests/test_gpu_streaming.py tests/test_gpu_kernels.py -m gpu -q -x -k "generate or voice or continuous or pretrained or streamer or streaming or pcm" > $O/pytest.log 2>&1); grep -E "passed|failed|error" $O/pytest.log | tail -3
