#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04g; mkdir -p $O
cd $R; export TMPDIR=/tmp
for so in liblc.so; do
for a in "1.5b 10" "7b 20"; do
  n=$(echo $a | tr ' ' '_')_$so
  LC_SO=$so timeout 180 python tools/experiments/loader_consumer/bench_lc.py $a > $O/lc_$n.json 2> $O/lc_$n.err
  echo "== $so $a rc=$?"; python -c "
import json;d=json.load(open('$O/lc_$n.json'))
for k in ('persistent_vs_torch_after_1_step','abort_word','persistent_over_launch_chain'): print(k, d.get(k))
print({k:v for k,v in d['timeline']['cu0'].items()})
print('us/layer launches', (d.get('launch_chain') or {}).get('us_per_layer'), 'persistent', (d.get('persistent') or {}).get('us_per_layer'), 'GB/s', (d.get('launch_chain') or {}).get('GBps'), (d.get('persistent') or {}).get('GBps'))"; tail -2 $O/lc_$n.err | grep -v amdgpu.ids
done; done
