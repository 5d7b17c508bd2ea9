#!/bin/bash
# last seconds of the round's GPU budget: does the number of HIP hardware queues (GPU_MAX_HW_QUEUES, default 4) explain what the
# data-parallel path (6-7 streams) loses against the plain one (4 streams)?  one-rank data-parallel launch line, 8 queues vs default
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
A="--gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extra"
for q in 8 4; do
  GPU_MAX_HW_QUEUES=$q timeout 40 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2956$q bench.py $A > $O/r2f7_dp1_q$q.json 2> $O/r2f7_dp1_q$q.err
  python -c "
import json; d=json.loads([l for l in open('$O/r2f7_dp1_q$q.json') if l.startswith('{')][-1]); print('queues $q', d['value'], d['ms_per_step'])"
done
