#!/bin/bash
# round 4, call k: pipelines (text through the pipeline's own encoder, inpainting) + the whole sampling file
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sampling.py -q -x 2>&1 | tail -25 > gpurun_out/r4_k_pytest.txt
cat gpurun_out/r4_k_pytest.txt
