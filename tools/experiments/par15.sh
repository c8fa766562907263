# The GPU-dependent 1.5B parity NaN, paired on one GPU: the default line (without its configs[3] leg) with the library as it was BEFORE the
# memset / memcpy nodes were replaced by kernels (build/variants/libvvhip_memsetnodes.so = commit ad5f27a's sources), then with the library
# at HEAD, then the sampler's stage probe resetting its records through a memset node (VVHIP_NAN_PROBE=2) and through a kernel (=1)
cd ${GRAFT_REPO_ROOT:-.}; O=gpurun_out/${1:-r06z}; mkdir -p $O
A="--no-config3"
run() { env $2 timeout 600 python bench.py $A $3 > $O/$1.json 2> $O/$1.err; }
chk() { python - <<PY
import json,sys
try:
    d=json.load(open("$O/$1.json")); p=d["extra"]["configs"]["configs[1]"]["parity"]
    print("$1", d["extra"]["libvvhip_build_id"], d["ms_per_step"], "main", d["parity"]["within_bounds"], "1.5B", p["within_bounds"], p["vs_fp32"]["latent"], p["vs_fp32"]["nonfinite_steps"])
    sys.exit(0 if not p["vs_fp32"]["nonfinite_steps"] else 7)
except Exception as e:
    print("$1 ERR", repr(e)[:200]); sys.exit(1)
PY
}
rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id:" | head -1
run before "VVHIP_LIB=build/variants/libvvhip_memsetnodes.so VVHIP_ALLOW_FOREIGN_NODES=1"; chk before; rc=$?
run head ""; chk head
if [ $rc -eq 7 ]; then echo "this GPU shows the NaN with the memset-node library"; run head2 ""; chk head2; fi
if [ "$2" = "probe" ]; then
  VVHIP_NAN_PROBE=2 POISON_HBM=0 VVHIP_POISON=0 timeout 600 python tools/experiments/poison_hunt.py 1p5b > $O/probe2.out 2> $O/probe2.err; grep -a "\[main\]" $O/probe2.out
  echo "memset-node reset: calls whose record buffer holds words nobody wrote: $(grep -ac 'nobody wrote' $O/probe2.err)"; grep -a "nobody wrote" $O/probe2.err | head -3 | cut -c1-250
  VVHIP_NAN_PROBE=1 POISON_HBM=0 VVHIP_POISON=0 timeout 600 python tools/experiments/poison_hunt.py 1p5b > $O/probe1.out 2> $O/probe1.err; grep -a "\[main\]" $O/probe1.out
  echo "kernel reset: calls whose record buffer holds words nobody wrote: $(grep -ac 'nobody wrote' $O/probe1.err); calls with non-finite stages: $(grep -ac 'stages hold' $O/probe1.err)"
fi
