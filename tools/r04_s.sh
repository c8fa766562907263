#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04s; mkdir -p $O
cd $R; export TMPDIR=/tmp
timeout 200 python tools/experiments/staggered_branches/bench_stagger.py 7b 20 > $O/stagger_7b.json 2> $O/stagger_7b.err; tail -3 $O/stagger_7b.err | cut -c1-300
timeout 200 python tools/experiments/staggered_branches/bench_stagger.py 1.5b 10 > $O/stagger_1p5b.json 2> $O/stagger_1p5b.err; tail -3 $O/stagger_1p5b.err | cut -c1-300
for f in $O/stagger_*.json; do python -c "
import json;d=json.load(open('$f'));print(d['shape'], {k:(v['us_per_layer'], v['GBps']) for k,v in d.items() if isinstance(v,dict)}, {k:v for k,v in d.items() if k.endswith('over_chain')})"; done
