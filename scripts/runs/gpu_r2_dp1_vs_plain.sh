#!/bin/bash
# same box: the plain single-GPU launch against the driver's data-parallel launch line at one rank (what the reducer path itself costs)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
A="--steps 10 --warmup 3 --no-cpu-baseline --no-extra"
timeout 100 python bench.py $A > $O/r2f5_plain.json 2> $O/r2f5_plain.err
timeout 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 $A > $O/r2f5_dp1.json 2> $O/r2f5_dp1.err
for n in plain dp1; do python -c "
import json; d=json.loads([l for l in open('$O/r2f5_$n.json') if l.startswith('{')][-1]); print('$n', d['value'], d['ms_per_step'])"; done
