# A/B on one box: one utterance's decode -> semantic re-encode as ONE captured sequence (vv_codec_chain_batch with n = 1) against two
# (vv_codec_decode + vv_semantic_encode); arms alternating
cd ${GRAFT_REPO_ROOT:-.}; O=gpurun_out/${1:-r06_chain_single}; mkdir -p $O
Q="--no-cpu-baseline --no-eager-baseline --skip-extra --no-parity --no-config3 --no-roofline"
for rep in 1 2 3; do for arm in 0 1; do
  VVHIP_CHAIN_SINGLE=$arm timeout 300 python bench.py $Q --steps 100 --warmup 10 > $O/7b_${arm}_$rep.json 2> $O/7b_${arm}_$rep.err
  VVHIP_CHAIN_SINGLE=$arm timeout 300 python bench.py $Q --workload 1p5b --steps 200 --warmup 10 > $O/1p5b_${arm}_$rep.json 2> $O/1p5b_${arm}_$rep.err
  python - <<PY
import json
a=json.load(open("$O/7b_${arm}_$rep.json")); b=json.load(open("$O/1p5b_${arm}_$rep.json"))
print("chain_single=$arm rep $rep: 7B", a["ms_per_step"], "ms; 1.5B", b["ms_per_step"], "ms")
PY
done; done
