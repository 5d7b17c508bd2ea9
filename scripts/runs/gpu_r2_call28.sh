#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for rep in 1 2; do for v in 0 1 2 3; do echo "bf16_operands=$v: $(MUSE_UVIT_BF16_OPERANDS=$v timeout 300 python scripts/uvit_bench.py 64 3 bf16 256 adamw 2>&1 | tail -1 | cut -c40-140)"; done; done
echo "wgrad_stream=0 ops=3: $(MUSE_WGRAD_STREAM=0 timeout 300 python scripts/uvit_bench.py 64 3 bf16 256 adamw 2>&1 | tail -1 | cut -c40-140)"
