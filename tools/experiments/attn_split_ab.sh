# A/B on one box: positions per decode-attention split for the north-star step (one 32K row: 1024 -> 32 splits x 4 kv heads = 128 workgroups)
cd ${GRAFT_REPO_ROOT:-.}; O=gpurun_out/${1:-r06_attn_split}; mkdir -p $O
Q="--no-cpu-baseline --no-eager-baseline --skip-extra --no-parity --no-config3 --steps 60 --warmup 10"
for rep in 1 2; do for sp in 1024 512 256; do
  VVHIP_ATTN_SPLIT_POS=$sp timeout 300 python bench.py $Q > $O/sp${sp}_$rep.json 2> $O/sp${sp}_$rep.err
  python - <<PY
import json
d=json.load(open("$O/sp${sp}_$rep.json")); a=(d["roofline"] or {}).get("attention") or {}
print("split_pos $sp rep $rep: ms_per_step", d["ms_per_step"], "attention unit us", a.get("avg_launch_us"), "frac", a.get("frac"))
PY
done; done
