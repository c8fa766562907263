#!/bin/bash
# skewed v3 prefill attention (row half 1 runs its P.V one stage late): row-wise parity, timed-mode suite, prefill A/B, default line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/z; mkdir -p $O
cd $R; export TMPDIR=/tmp
(time timeout 400 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_timed_mode.py -x -q -m gpu -k "prefill or timed or real_widths") > $O/pytest_skew.log 2>&1; tail -6 $O/pytest_skew.log
Q="--skip-extra --no-cpu-baseline --no-roofline --steps 6 --warmup 2"
for v in default noskew; do
  case $v in default) E="";; noskew) E="VVHIP_ATTN3_NOSKEW=1";; esac
  env $E timeout 300 python bench.py $Q > $O/ab_$v.json 2>$O/ab_$v.err
  echo $v $(python -c "import json; d=json.load(open('$O/ab_$v.json')); print(d['ms_per_step'], d['extra']['prefill_phases'])" 2>&1 | tail -1)
done
(time timeout 500 python bench.py --steps 20 --warmup 5) > $O/r02_bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.err
python -c "import json; d=json.load(open('$O/r02_bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['gpu_eager_baseline'], d['extra']['prefill_phases'])"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $O/p_mfma -o m -- python bench.py --skip-extra --no-cpu-baseline --no-roofline --steps 2 --warmup 1 > /dev/null 2> $O/pmc_mfma.err
python tools/rocprof_summary.py $O/p_mfma/m_results.db $O/r02_7b_pmc_mfma --pmc > $O/r02_7b_pmc_mfma_top.txt 2>&1; rm -rf $O/p_mfma
grep -E "attn_prefill" $O/r02_7b_pmc_mfma_top.txt | cut -c1-170 | head -4
