#!/bin/bash
# round 4, call o: gradient accumulation under the reducer (no_sync) on an RCCL group of one rank
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_uvit.py -q -x -k "accumulation or reduced_inside_backward" 2>&1 | grep -v amdgpu.ids | tail -25 > gpurun_out/r4_o_pytest.txt
cat gpurun_out/r4_o_pytest.txt
