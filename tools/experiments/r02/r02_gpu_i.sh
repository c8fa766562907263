#!/bin/bash
# session I: decode-attention probe (one 7B-geometry layer, 32K positions): splits x waves x load policy
mkdir -p gpurun_out
O=gpurun_out
export TMPDIR=/tmp
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d $O/pa_$name -o p -- python tools/probe_attn.py --geom 7b --len 32000 --rows ${ROWS:-1} --steps 200 > $O/pa_$name.txt 2> $O/pa_$name.err
  python tools/rocprof_summary.py $O/pa_$name/p_results.db $O/pa_$name > /dev/null 2>&1
  rm -rf $O/pa_$name
  echo "$name: $(tail -1 $O/pa_$name.txt | cut -c1-90) | $(grep attn_fused $O/pa_${name}_kernel_shapes.csv | head -2 | tr '\n' ' ' | cut -c1-220)"
}
run s1024_w4 VVHIP_ATTN_SPLIT_POS=1024
run s512_w4 VVHIP_ATTN_SPLIT_POS=512
run s2048_w4 VVHIP_ATTN_SPLIT_POS=2048
run s1024_w8 VVHIP_ATTN_SPLIT_POS=1024 VVHIP_ATTN_LONG=4096
run s512_w8 VVHIP_ATTN_SPLIT_POS=512 VVHIP_ATTN_LONG=4096
run s1024_w4_nt VVHIP_ATTN_SPLIT_POS=1024 VVHIP_LIB=build/variants/libvvhip_nt.so
run s512_w4_nt VVHIP_ATTN_SPLIT_POS=512 VVHIP_LIB=build/variants/libvvhip_nt.so
run s1024_w4_contig VVHIP_ATTN_SPLIT_POS=1024 VVHIP_ATTN_CONTIGUOUS=1
ROWS=8 run b8_s1024_w4 VVHIP_ATTN_SPLIT_POS=1024
ROWS=8 run b8_s1024_w4_nt VVHIP_ATTN_SPLIT_POS=1024 VVHIP_LIB=build/variants/libvvhip_nt.so
ROWS=8 run b8_s4096_w4 VVHIP_ATTN_SPLIT_POS=4096
