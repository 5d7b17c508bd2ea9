#!/bin/bash
# taming VQGANModel: stride-2 conv gather, model vs reference goldens / oracle; existing conv + VQGAN tests for regressions
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu -s -k "taming or vqgan or vq_indices" 2>&1 | tail -15
