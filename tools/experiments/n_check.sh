cd ${GRAFT_REPO_ROOT:-.}; O=gpurun_out/r06_head4; mkdir -p $O
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_torchrun_n1.json 2> $O/t1.err; echo "torchrun n1 rc $?"
VVHIP_BENCH_SHARED_GPU=1 timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-eager-baseline > $O/r06_shared_gpu_n2_controlflow.json 2> $O/d2.err; echo "shared n2 rc $?"
python - <<PY
import json
for f in ("r06_torchrun_n1","r06_shared_gpu_n2_controlflow"):
    try:
        d=json.load(open("$O/%s.json"%f)); c3=(d["extra"].get("configs") or {}).get("configs[3] per GPU") or {}
        print(f, d["n_gpus"], d["value"], d["ms_per_step"], "parity", (d.get("parity") or {}).get("within_bounds"), "c3", c3.get("value"), c3.get("ms_per_step"), "rccl" in d["extra"] and bool(d["extra"]["rccl"]))
    except Exception as e: print(f, "ERR", repr(e)[:200])
PY
tail -n 3 $O/t1.err $O/d2.err | cut -c1-200
