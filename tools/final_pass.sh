R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_head; mkdir -p $O; cd $R; export TMPDIR=/tmp
Q="--no-cpu-baseline --no-eager-baseline --skip-extra --no-parity"
timeout 900 python bench.py > $O/r06_bench_default.json 2> $O/b0.err
timeout 300 python bench.py --workload 1p5b --steps 60 --warmup 10 $Q > $O/r06_1p5b.json 2> $O/e1.err
timeout 300 python bench.py --workload streaming --steps 60 > $O/r06_streaming.json 2> $O/e2.err
timeout 600 python bench.py --batch 8 --speakers 4 --text-tokens 10569 --steps 20 --warmup 5 $Q > $O/r06_7b_4spk_batch8_32k.json 2> $O/e3.err
bash tools/pmc_refresh.sh r06 $O | tail -5
cp profiles/pmc_traffic.json $O/pmc_traffic.json
cp gpurun_out/pmc/r06_*_pmc_fetch_* $O/ 2>/dev/null
timeout 900 python bench.py > $O/r06_bench_default_with_traffic.json 2> $O/b1.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_stats -o s -- python bench.py --steps 20 --warmup 5 $Q --skip-extra --no-roofline > /dev/null 2> $O/rp_stats.err
find $O/p_stats -name "*kernel_stats*" -exec cp {} $O/r06_bench_default_rocprofv3_stats.csv \; ; rm -rf $O/p_stats
timeout 900 python bench.py --full-utterance > $O/r06_full_utterance.json 2> $O/utt.err
python - <<PY
import json
d=json.load(open("$O/r06_bench_default_with_traffic.json")); r=d["roofline"]
print("main", d["value"], d["ms_per_step"], r["frac"], "traffic", r["traffic"], r["bytes_per_launch"], "attn traffic", r["attention"]["traffic"])
c3=d["extra"]["configs"]["configs[3] per GPU"]; r3=c3["roofline"]
print("c3", c3["value"], c3["ms_per_step"], r3["frac"], "traffic", r3["traffic"], r3["bytes_per_launch"], r3["attention"]["traffic"], (r3.get("gemv_other") or {}).get("traffic"))
u=json.load(open("$O/r06_full_utterance.json")); print("utt", u["value"], u["extra"].get("utterance_wall_s"), u["extra"].get("utterance_audio_s"))
PY
head -5 $O/r06_bench_default_rocprofv3_stats.csv | cut -c1-200
