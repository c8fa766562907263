cd ${GRAFT_REPO_ROOT:-.}; O=gpurun_out/${1:-r06y}; mkdir -p $O
rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id:" | head -1
timeout 600 python tools/experiments/poison_hunt.py 1p5b > $O/poison.out 2> $O/poison.err; echo "poison rc $?"; grep -a "\[main\]\|\[poison\]" $O/poison.out; grep -a "nan_probe\|Error\|error" $O/poison.err | head -20
VVHIP_NAN_PROBE=1 POISON_HBM=0 VVHIP_POISON=0 timeout 600 python tools/experiments/poison_hunt.py 1p5b > $O/probe.out 2> $O/probe.err; echo "probe rc $?"; grep -a "\[main\]" $O/probe.out
grep -a "nobody wrote\|sync " $O/probe.err | head -12; grep -a "nan_probe\] samp" $O/probe.err | awk '{print $3,$4,$5,$6,$7,$8}' | uniq -c | head -30
