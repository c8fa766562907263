#!/bin/bash
# round-2 GPU session C: parity suite, north-star bench (one-chunk prefill, new attention), decode-attention sweep, tokenizer A/B,
# batch-8 @32K, rocprof + PMC (summarised here: the rocpd databases are too big to travel back)
mkdir -p gpurun_out
O=gpurun_out
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -150) > $O/r02c_pytest.log 2>&1
(time timeout 600 python bench.py --steps 20 --warmup 5) > $O/r02c_bench.json 2> $O/r02c_bench.err
NS="--skip-extra --no-cpu-baseline --no-roofline --steps 20 --warmup 5"
# decode attention at 32K: positions per split x waves per workgroup
for cfg in "512 4096" "1024 4096" "2048 4096" "512 100000000" "1024 100000000"; do
  set -- $cfg
  VVHIP_ATTN_SPLIT_POS=$1 VVHIP_ATTN_LONG=$2 timeout 300 python bench.py $NS > $O/r02c_ns_attn_$1_$2.json 2>/dev/null
done
# prefill in 2048-row chunks (round-1 chunking) for comparison with the one-chunk default
timeout 300 python bench.py $NS --prefill-rows 2048 > $O/r02c_ns_chunk2048.json 2>/dev/null
# tokenizer-chain A/B on the 1.5B workload
T15="--workload 1p5b --steps 150 --warmup 10 --no-cpu-baseline --no-roofline"
timeout 300 python bench.py $T15 > $O/r02c_1p5b_new.json 2>/dev/null
VVHIP_NO_ROWS_NORMDW=1 timeout 300 python bench.py $T15 > $O/r02c_1p5b_norows.json 2>/dev/null
VVHIP_NO_CONV_KERNELS=1 timeout 300 python bench.py $T15 > $O/r02c_1p5b_noconv.json 2>/dev/null
VVHIP_NO_ROWS_NORMDW=1 VVHIP_NO_CONV_KERNELS=1 timeout 300 python bench.py $T15 > $O/r02c_1p5b_oldtok.json 2>/dev/null
# BASELINE configs[3] on one GPU: 7B, 4 speakers, 8 utterances, each at 32K context, N = 20
timeout 600 python bench.py --batch 8 --speakers 4 --text-tokens 10569 --steps 20 --warmup 5 --skip-extra --no-cpu-baseline > $O/r02c_7b_4spk_batch8_32k.json 2> $O/r02c_b8.err
# rocprof
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r02c_prof -o ns -- python bench.py --skip-extra --no-cpu-baseline --steps 20 --warmup 5 > $O/r02c_bench_rocprof.json 2> $O/r02c_rocprof.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $O/r02c_pmc_mfma -o mfma -- python bench.py --skip-extra --no-cpu-baseline --no-roofline --steps 2 --warmup 1 > $O/r02c_pmc_mfma.json 2> $O/r02c_pmc_mfma.err
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/r02c_pmc_fetch -o fetch -- python bench.py --skip-extra --no-cpu-baseline --no-roofline --steps 40 --warmup 2 > $O/r02c_pmc_fetch.json 2> $O/r02c_pmc_fetch.err
python tools/rocprof_summary.py $O/r02c_prof/ns_results.db $O/r02c_7b_northstar > $O/r02c_7b_northstar_top.txt 2>&1
python tools/rocprof_summary.py $O/r02c_pmc_mfma/mfma_results.db $O/r02c_7b_pmc_mfma --pmc > $O/r02c_7b_pmc_mfma_top.txt 2>&1
python tools/rocprof_summary.py $O/r02c_pmc_fetch/fetch_results.db $O/r02c_7b_pmc_fetch --pmc > $O/r02c_7b_pmc_fetch_top.txt 2>&1
rm -rf $O/r02c_prof $O/r02c_pmc_mfma $O/r02c_pmc_fetch
du -sh $O
tail -6 $O/r02c_pytest.log
head -c 600 $O/r02c_bench.json
