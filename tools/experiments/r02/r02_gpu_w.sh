#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_generate.py tests/test_gpu_timed_mode.py tests/test_gpu_fullsize.py tests/test_gpu_streaming.py -x -q -m gpu 2>&1 | grep -v "^$" | tail -4
NS="--skip-extra --no-cpu-baseline --no-roofline --steps 10 --warmup 3"
timeout 600 python bench.py $NS > $O/r02w_ns.json 2> $O/r02w_err.txt
python -c "import json;d=json.load(open('$O/r02w_ns.json'));print(d['ms_per_step'], d['extra']['prefill_phases'], d['extra'].get('prefill_tflops'))"
timeout 300 python bench.py --workload 1p5b --steps 100 --warmup 10 --no-cpu-baseline --no-roofline > $O/r02w_15.json 2>/dev/null
python -c "import json;d=json.load(open('$O/r02w_15.json'));print(d['ms_per_step'], d['extra']['prefill_phases'])"
