# where does a replayed memset node take its fill pattern from?  The engine's probe (VVHIP_NAN_PROBE=2: record buffer reset through
# hipMemsetAsync inside the captured sampler sequence) under three settings of the process
cd ${GRAFT_REPO_ROOT:-.}; O=gpurun_out/${1:-r06m}; mkdir -p $O
arm() { env POISON_HBM=0 VVHIP_POISON=0 VVHIP_NAN_PROBE=2 $2 timeout 600 python tools/experiments/poison_hunt.py 1p5b > $O/$1.out 2> $O/$1.err
        echo "$1 [$2]: $(grep -a '\[main\]' $O/$1.out | cut -c1-90) | calls with a stale fill: $(grep -ac 'nobody wrote' $O/$1.err)"; grep -a "nobody wrote" $O/$1.err | head -4 | sed 's/.*nobody wrote: /      /' ; }
arm plain ""
arm nocapture "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
arm perturb "MALLOC_PERTURB_=165"
arm devkernarg0 "HIP_FORCE_DEV_KERNARG=0"
