#!/bin/bash
# session J: deferred (separate-kernel) split merge: parity + probe + north-star / batch-8
mkdir -p gpurun_out
O=gpurun_out
export TMPDIR=/tmp
(time timeout 600 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_kernels.py tests/test_gpu_generate.py tests/test_gpu_streaming.py -m gpu -q --tb=short 2>&1 | tail -30) > $O/r02j_pytest.log 2>&1
run() {
  name=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d $O/pa_$name -o p -- python tools/probe_attn.py --geom ${GEOM:-7b} --len ${LEN:-32000} --rows ${ROWS:-1} --steps 200 > $O/pa_$name.txt 2> $O/pa_$name.err
  python tools/rocprof_summary.py $O/pa_$name/p_results.db $O/pa_$name > /dev/null 2>&1
  rm -rf $O/pa_$name
  echo "$name: $(tail -1 $O/pa_$name.txt | cut -c40-90) | $(grep -E 'attn_fused|merge2' $O/pa_${name}_kernel_shapes.csv | head -3 | cut -c1-40,100-160 | tr '\n' ' ')"
}
run defer_s1024 VVHIP_ATTN_SPLIT_POS=1024
run defer_s512 VVHIP_ATTN_SPLIT_POS=512
run defer_s256 VVHIP_ATTN_SPLIT_POS=256
run ticket_s1024 VVHIP_ATTN_SPLIT_POS=1024 VVHIP_ATTN_TICKET_MERGE=1
run defer_s512_contig VVHIP_ATTN_SPLIT_POS=512 VVHIP_ATTN_CONTIGUOUS=1
ROWS=8 run b8_defer VVHIP_ATTN_SPLIT_POS=1024
ROWS=8 run b8_defer_t512 VVHIP_ATTN_SPLIT_POS=1024 VVHIP_ATTN_TARGET_WGS=512
GEOM=1.5b LEN=64000 run g15_defer_s1024 VVHIP_ATTN_SPLIT_POS=1024
GEOM=1.5b LEN=64000 run g15_defer_s512 VVHIP_ATTN_SPLIT_POS=512
NS="--skip-extra --no-cpu-baseline --no-roofline --steps 20 --warmup 5"
for sp in 512 1024; do VVHIP_ATTN_SPLIT_POS=$sp timeout 300 python bench.py $NS > $O/r02j_ns_$sp.json 2>/dev/null; done
B8="--batch 8 --speakers 4 --text-tokens 10569 --steps 20 --warmup 5 --skip-extra --no-cpu-baseline --no-roofline"
timeout 600 python bench.py $B8 > $O/r02j_b8.json 2>/dev/null
for f in $O/r02j_*.json; do echo $(basename $f) $(python -c "import json;d=json.load(open('$f'));print(d['ms_per_step'])"); done
grep -E "passed|failed" $O/r02j_pytest.log
