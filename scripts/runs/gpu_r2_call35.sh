#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_models.py -q -m gpu -p no:cacheprovider -k "streams or train_step" 2>&1 | grep -E "passed|failed"
for b in 32 64; do timeout 400 python scripts/uvit_bench.py $b 2 bf16 1024 adamw 2>&1 | tail -1; done
timeout 400 python scripts/uvit_bench.py 128 2 bf16 256 adamw 2>&1 | tail -1
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
