#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "layernorm or ffn_mid" --tb=line 2>&1 | grep -E "passed|failed|Error" | head
timeout 600 python -m pytest tests/test_gpu_models.py -q -m gpu -p no:cacheprovider -x > $O/r2_call44_pytest.txt 2>&1; grep -E "passed|failed" $O/r2_call44_pytest.txt | tail -2
for rep in 1 2; do for v in 0 1 3; do
MUSE_LN_PAIR=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); hb=d['roofline']['hbm_bound_kernels']; print('ln_pair=$v', d['value'], d['ms_per_step'], 'ln_bwd', hb['layernorm_bwd']['ms_total'], 'ln_fwd', hb['layernorm_fwd']['ms_total'], 'tr_ms', d['extra']['transformer_fwd_bwd_ms'])"
done; done
