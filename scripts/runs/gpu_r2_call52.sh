#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "train_step or streams or transformer or layernorm or ffn_mid or reducer" > $O/r2_call52_pytest.txt 2>&1; grep -E "passed|failed|Error" $O/r2_call52_pytest.txt | tail -3
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
