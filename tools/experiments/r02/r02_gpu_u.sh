#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^$" | tail -6 > $O/r02u_tests.txt
cat $O/r02u_tests.txt
NS="--skip-extra --no-cpu-baseline --no-roofline --steps 20 --warmup 5"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p_ns -o ns -- python bench.py $NS > $O/r02u_ns.json 2> $O/r02u_err.txt
python tools/rocprof_summary.py $O/p_ns/ns_results.db $O/r02u_ns > $O/r02u_ns_top.txt 2>&1; rm -rf $O/p_ns
python -c "import json;d=json.load(open('$O/r02u_ns.json'));print(d['ms_per_step'], d['extra']['prefill_phases'])"
cat $O/r02u_ns_timeline.txt
