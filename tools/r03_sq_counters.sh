R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sq; mkdir -p $O; cd $R; export TMPDIR=/tmp
Q="--skip-extra --no-cpu-baseline --no-eager-baseline --no-roofline --steps 2 --warmup 1"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $O/p1 -o m -- python bench.py $Q > /dev/null 2> $O/pmc1.err
python tools/rocprof_summary.py $O/p1/m_results.db $O/r03_prefill_sq --pmc > $O/r03_prefill_sq_top.txt 2>&1; rm -rf $O/p1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/p2 -o m -- python bench.py $Q > /dev/null 2> $O/pmc2.err
python tools/rocprof_summary.py $O/p2/m_results.db $O/r03_prefill_sq2 --pmc > $O/r03_prefill_sq2_top.txt 2>&1; rm -rf $O/p2
ls $O
