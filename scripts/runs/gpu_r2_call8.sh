#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_uvit.py -q --tb=short -p no:cacheprovider -k "sampl or generate or mask or cond_dropout" > $O/r2_sampling_tests.txt 2>&1
tail -40 $O/r2_sampling_tests.txt | cut -c1-220
