#!/bin/bash
# round 4, call r: half-precision conditioning inputs inside autocast
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_uvit.py -q -x -k "half_precision_conditioning" 2>&1 | grep -v "amdgpu.ids" | tail -30 > gpurun_out/r4_r_pytest.txt
cat gpurun_out/r4_r_pytest.txt
