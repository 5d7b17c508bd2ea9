#!/bin/bash
# generate2 under a HIP graph, pipeline with the taming tokenizer
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_sampling.py -x -q -m gpu -s -k "generate2" 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "pipeline" 2>&1 | tail -8
