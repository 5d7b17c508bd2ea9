#!/bin/bash
# re-run of the one test that failed in gpu_r2_final2.sh on a missing .detach(), and the data-parallel launch line at one rank with the
# AdamW-behind-each-bucket update on / off
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_models.py -m gpu -q --tb=short -rP -p no:cacheprovider -k "full_depth_vs_reference_golden or vqgan_f16_256_vs_reference_golden" > $O/r2f3_pytest.txt 2>&1; echo "pytest exit $?" >> $O/r2f3_pytest.txt
grep -E "passed|failed|pytest exit|vs the reference" $O/r2f3_pytest.txt | cut -c1-600
for v in 1 0; do
  MUSE_OPT_IN_REDUCER=$v timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$v bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $O/r2f3_dp1_optred$v.json 2> $O/r2f3_dp1_optred$v.err
  python -c "
import json; d=json.loads([l for l in open('$O/r2f3_dp1_optred$v.json') if l.startswith('{')][-1]); print('opt_in_reducer=$v', d['value'], d['ms_per_step'])"
done
