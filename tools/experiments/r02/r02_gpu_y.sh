#!/bin/bash
# v3 prefill attention: row-wise parity (permlane vs LDS-permute exchange vs v2 vs oracle), the xsplit=1 suites that prefill
# through it, A/B of the 7B prompt prefill, the default bench line with the PyTorch-ROCm eager leg, rocprof kernel trace.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/y; mkdir -p $O
cd $R; export TMPDIR=/tmp
(time timeout 500 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_timed_mode.py tests/test_gpu_fullsize.py -x -q -m gpu) > $O/pytest_v3.log 2>&1; tail -15 $O/pytest_v3.log
Q="--skip-extra --no-cpu-baseline --no-roofline --steps 6 --warmup 2"
for v in default shfl v2; do
  case $v in default) E="";; shfl) E="VVHIP_ATTN3_SHFL=1";; v2) E="VVHIP_ATTN3=0";; esac
  env $E timeout 300 python bench.py $Q > $O/ab_$v.json 2>$O/ab_$v.err
  echo $v $(python -c "import json; d=json.load(open('$O/ab_$v.json')); print(d['ms_per_step'], d['extra']['prefill_phases'])" 2>&1 | tail -1)
done
(time timeout 500 python bench.py --steps 20 --warmup 5) > $O/r02_bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.err
python -c "import json; d=json.load(open('$O/r02_bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline'], d['gpu_eager_baseline'], d['extra']['prefill_phases'])"
NS="--skip-extra --no-cpu-baseline --steps 20 --warmup 5"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/p_ns -o ns -- python bench.py $NS > $O/r02_bench_under_rocprof.json 2> $O/rocprof_ns.err
python tools/rocprof_summary.py $O/p_ns/ns_results.db $O/r02_7b_northstar --around vv_attn_prefill3 40 > $O/r02_7b_northstar_top.txt 2>&1; rm -rf $O/p_ns
head -12 $O/r02_7b_northstar_top.txt | cut -c1-150
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $O/p_mfma -o m -- python bench.py --skip-extra --no-cpu-baseline --no-roofline --steps 2 --warmup 1 > /dev/null 2> $O/pmc_mfma.err
python tools/rocprof_summary.py $O/p_mfma/m_results.db $O/r02_7b_pmc_mfma --pmc > $O/r02_7b_pmc_mfma_top.txt 2>&1; rm -rf $O/p_mfma
grep -E "attn_prefill|gemm4" $O/r02_7b_pmc_mfma_top.txt | cut -c1-170 | head -12
du -sh $O
