#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "ffn_mid" 2>&1 | grep -E "passed|failed|Error|assert" | head
timeout 600 python -m pytest tests/test_gpu_models.py -q -m gpu -p no:cacheprovider -x > $O/r2_call42_pytest.txt 2>&1; grep -E "passed|failed" $O/r2_call42_pytest.txt | tail -2
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); hb=d['roofline']['hbm_bound_kernels']; print(d['value'], d['ms_per_step'], 'ffn_bwd', hb['ffn_mid_bwd'], 'ffn_fwd', hb['ffn_mid_fwd'], 'tr_ms', d['extra']['transformer_fwd_bwd_ms'])"
timeout 200 python bench.py --uvit-leg 64,256,2 2>/dev/null | tail -1
