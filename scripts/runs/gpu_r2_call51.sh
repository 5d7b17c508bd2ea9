#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_models.py -q -m gpu -p no:cacheprovider -x > $O/r2_call51_pytest.txt 2>&1; grep -E "passed|failed|Error" $O/r2_call51_pytest.txt | tail -3
for rep in 1 2; do for v in 0 1; do
MUSE_OPT_IN_BACKWARD=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('opt_in_backward=$v', d['value'], d['ms_per_step'])"
done; done
