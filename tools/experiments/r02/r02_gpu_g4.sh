#!/bin/bash
# gemm4 with the peeled straight-line stage (fragment requests first, LDS-DMA issue between the two k-tiles' MFMAs): parity + prefill time
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/g4; mkdir -p $O
cd $R; export TMPDIR=/tmp
(time timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_geometry.py -x -q -m gpu -k "gemm3 or prefill or real_widths") > $O/pytest.log 2>&1; tail -4 $O/pytest.log
Q="--skip-extra --no-cpu-baseline --no-roofline --steps 6 --warmup 2"
timeout 300 python bench.py $Q > $O/ab_default.json 2>$O/ab_default.err
python -c "import json; d=json.load(open('$O/ab_default.json')); print(d['ms_per_step'], d['extra']['prefill_phases'], d['extra'].get('prefill_tflops'))"
B="python bench.py --skip-extra --no-cpu-baseline --no-roofline --steps 2 --warmup 1"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $O/p1 -o m -- $B > /dev/null 2> $O/pmc1.err
python tools/rocprof_summary.py $O/p1/m_results.db $O/sq1 --pmc > $O/sq1_top.txt 2>&1; rm -rf $O/p1
grep -hE "gemm4" $O/sq1_top.txt | cut -c1-30,92-170
