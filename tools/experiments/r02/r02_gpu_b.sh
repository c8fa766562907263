#!/bin/bash
# round-2 GPU session B: parity suite, north-star bench, batch-8 @32K, continuous admission, A/B switches, rocprof + PMC
mkdir -p gpurun_out
O=gpurun_out
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -150) > $O/r02b_pytest.log 2>&1
(time timeout 600 python bench.py --steps 20 --warmup 5) > $O/r02b_bench.json 2> $O/r02b_bench.err
# prefill A/B on the same box: old tile GEMM / old prefill attention
VVHIP_NO_TILE3=1 VVHIP_NO_ATTN2=1 timeout 300 python bench.py --steps 5 --warmup 2 --skip-extra --no-cpu-baseline --no-roofline > $O/r02b_ns_oldprefill.json 2>/dev/null
VVHIP_NO_TILE3=1 timeout 300 python bench.py --steps 5 --warmup 2 --skip-extra --no-cpu-baseline --no-roofline > $O/r02b_ns_oldgemm_newattn.json 2>/dev/null
# tokenizer-chain A/B on the 1.5B workload
timeout 300 python bench.py --workload 1p5b --steps 150 --warmup 10 --no-cpu-baseline > $O/r02b_1p5b_new.json 2>/dev/null
VVHIP_NO_ROWS_NORMDW=1 VVHIP_NO_CONV_KERNELS=1 timeout 300 python bench.py --workload 1p5b --steps 150 --warmup 10 --no-cpu-baseline > $O/r02b_1p5b_oldtok.json 2>/dev/null
# BASELINE configs[3] on one GPU: 7B, 4 speakers, 8 utterances, each at 32K context, N = 20
timeout 600 python bench.py --batch 8 --speakers 4 --text-tokens 10569 --steps 20 --warmup 5 --skip-extra --no-cpu-baseline > $O/r02b_7b_4spk_batch8_32k.json 2> $O/r02b_b8.err
# continuous admission at real shapes: 12 utterances through 4 slots (1.5B)
timeout 300 python bench.py --workload 1p5b --batch 4 --continuous 12 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/r02b_1p5b_continuous.json 2> $O/r02b_cont.err
# rocprof: kernel trace of the north-star command, then PMC passes (separate runs, kernel-trace only)
cd /tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r02b_prof -o ns -- python bench.py --skip-extra --no-cpu-baseline --steps 20 --warmup 5 > $O/r02b_bench_rocprof.json 2> $O/r02b_rocprof.err
rocprofv3 -L > $O/r02b_counters.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $O/r02b_pmc_mfma -o mfma -- python bench.py --skip-extra --no-cpu-baseline --no-roofline --steps 2 --warmup 1 > $O/r02b_pmc_mfma.json 2> $O/r02b_pmc_mfma.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/r02b_pmc_fetch -o fetch -- python bench.py --skip-extra --no-cpu-baseline --no-roofline --steps 2 --warmup 1 > $O/r02b_pmc_fetch.json 2> $O/r02b_pmc_fetch.err
# summarise the rocpd databases here (they are too big to travel back) and drop them
python tools/rocprof_summary.py $O/r02b_prof/ns_results.db $O/r02b_7b_northstar > $O/r02b_7b_northstar_top.txt 2>&1
python tools/rocprof_summary.py $O/r02b_pmc_mfma/mfma_results.db $O/r02b_7b_pmc_mfma --pmc > $O/r02b_7b_pmc_mfma_top.txt 2>&1
python tools/rocprof_summary.py $O/r02b_pmc_fetch/fetch_results.db $O/r02b_7b_pmc_fetch --pmc > $O/r02b_7b_pmc_fetch_top.txt 2>&1
rm -rf $O/r02b_prof $O/r02b_pmc_mfma $O/r02b_pmc_fetch
du -sh $O
tail -4 $O/r02b_pytest.log
head -c 1500 $O/r02b_bench.json
