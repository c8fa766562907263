cd ${GRAFT_REPO_ROOT:-.}; O=gpurun_out/r06v; mkdir -p $O
rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id:" | head -1
A="--workload 1p5b --steps 60 --warmup 10 --skip-extra --no-parity-long --cpu-frames 2"
run() { env $2 timeout 300 python bench.py $A $3 > $O/$1.json 2> $O/$1.err; python - <<PY
import json
try:
    d=json.load(open("$O/$1.json")); p=d["parity"]
    print("$1", d["ms_per_step"], p["within_bounds"], p["vs_fp32"]["latent"], p["vs_fp32"]["nonfinite_steps"], "eager", (d.get("gpu_eager_baseline") or {}).get("ms_per_step"))
except Exception as e: print("$1 ERR", e)
PY
}
run eager_on "" ""
run eager_off "" "--no-eager-baseline"
run eager_on_nograph "" "--no-graph"
run eager_on_noroof "" "--no-roofline"
run eager_on_tail0 "VVHIP_HEAD_TAIL=0" ""
