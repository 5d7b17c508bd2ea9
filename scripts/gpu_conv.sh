#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "conv or groupnorm" > $O/pytest_conv.txt 2>&1; tail -12 $O/pytest_conv.txt
WHICH=conv timeout 300 python scripts/gemm_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/probe_conv.txt
