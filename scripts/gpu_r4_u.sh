#!/bin/bash
# round 4, call u: the reference's offline-EMA and inpainting-log script flows
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ema.py tests/test_gpu_sampling.py -q -x -k "script_flow" 2>&1 | grep -v "amdgpu.ids\|Writing model\|Loading weights" | tail -30 > gpurun_out/r4_u_pytest.txt
cat gpurun_out/r4_u_pytest.txt
