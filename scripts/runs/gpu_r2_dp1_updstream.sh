#!/bin/bash
# data-parallel launch line at one rank: per-bucket AdamW on its own normal-priority stream against on the reducer's stream (same box),
# after the bit-identity test of the path
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
timeout 100 python -m pytest tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider -k "behind_each_reduced_bucket or torchrun_single_rank" > $O/r2f6_pytest.txt 2>&1; echo "pytest exit $?" >> $O/r2f6_pytest.txt
grep -E "passed|failed|pytest exit|^E " $O/r2f6_pytest.txt | head -8
A="--gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extra"
for v in own comm; do
  MUSE_OPT_REDUCER_STREAM=$v timeout 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py $A > $O/r2f6_dp1_$v.json 2> $O/r2f6_dp1_$v.err
  python -c "
import json; d=json.loads([l for l in open('$O/r2f6_dp1_$v.json') if l.startswith('{')][-1]); print('$v', d['value'], d['ms_per_step'])"
done
