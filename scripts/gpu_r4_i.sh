#!/bin/bash
# round 4, call i: force_down_up_sample (space-to-depth kernel, model vs the reference golden, modes, decoding) + the U-ViT file
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_uvit.py -q -x 2>&1 | tail -15 > gpurun_out/r4_i_pytest.txt
cat gpurun_out/r4_i_pytest.txt
