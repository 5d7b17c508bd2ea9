#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "layernorm_pair" --tb=short -s 2>&1 | grep -E "passed|failed|Error|assert|ln pair" | head
