#!/bin/bash
# the torchrun / RCCL path of bench.py on the one visible GPU: process-group init, packed-blob broadcast (2 GiB buckets), barrier +
# all-reduce around the timed region; and the self-spawning --gpus 2 request on a 1-GPU box (must fail with a clear message)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dist1; mkdir -p $O
cd $R; export TMPDIR=/tmp
(time timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --skip-extra --no-cpu-baseline) > $O/torchrun_n1.json 2> $O/torchrun_n1.err; tail -c 600 $O/torchrun_n1.err
python -c "import json; d=json.loads(open('$O/torchrun_n1.json').read().strip().splitlines()[-1]); print(d['n_gpus'], d['value'], d['ms_per_step'], d['extra']['weights_broadcast'], d['config']['parallelism'])"
timeout 120 python bench.py --gpus 2 --steps 2 > $O/gpus2.out 2> $O/gpus2.err; echo "rc=$?"; tail -2 $O/gpus2.err
