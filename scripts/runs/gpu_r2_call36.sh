#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "embedding" 2>&1 | grep -E "passed|failed|Error|error" | head -5
timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_uvit.py -q -m gpu -p no:cacheprovider -x > $O/r2_call36_pytest.txt 2>&1; grep -E "passed|failed" $O/r2_call36_pytest.txt | tail -2
for v in 0 1; do
MUSE_EMBED_BWD_SORT=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sort=$v', d['value'], d['ms_per_step'], 'tr_ms', d['extra']['transformer_fwd_bwd_ms'])"
done
