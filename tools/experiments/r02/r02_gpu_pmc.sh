#!/bin/bash
# where the prefill attention / GEMM waves spend their cycles: SQ counters (own --pmc passes, kernel-trace only)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc; mkdir -p $O
cd $R; export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1; grep -c "SQ_" $O/counters_list.txt
B="python bench.py --skip-extra --no-cpu-baseline --no-roofline --steps 2 --warmup 1"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o m -- $B > /dev/null 2> $O/pmc$i.err
  if [ -f $O/p$i/m_results.db ]; then python tools/rocprof_summary.py $O/p$i/m_results.db $O/sq$i --pmc > $O/sq${i}_top.txt 2>&1; else tail -3 $O/pmc$i.err; fi
  rm -rf $O/p$i
done
grep -hE "attn_prefill3|gemm4_kernel<3>" $O/sq*_top.txt | cut -c1-60,92-170
