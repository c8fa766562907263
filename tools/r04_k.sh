#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04k; mkdir -p $O
cd $R; export TMPDIR=/tmp
Q="--no-cpu-baseline --no-eager-baseline --skip-extra --no-roofline"
timeout 300 rocprofv3 --kernel-trace -d $O/p -o t -- python bench.py --workload 1p5b --steps 10 --warmup 3 $Q > $O/b.json 2> $O/rp.err
python tools/rocprof_summary.py $O/p/t_results.db $O/r04_1p5b_short > $O/top.txt 2>&1; rm -rf $O/p
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/r04_1p5b_short_kernel_shapes.csv")))
for r in rows:
    if any(k in r['Name'] for k in ('vv_gemm3','vv_gemm4','vv_g3_reduce','vv_pack_rows','vv_attn_prefill','vv_rope_append','vv_embed','vv_rmsnorm_rows','vv_gemm_tile','vv_stem','vv_block1d','at::')):
        print(f"{int(r['TotalDurationNs'])/1e3:9.1f} us {int(r['Calls']):5d} x {float(r['AverageNs'])/1e3:7.1f}  grid {int(r['GridX'])//int(r['WorkgroupX']):5d},{r['GridY']:>3}  {r['Name'][:60]}")
PY
