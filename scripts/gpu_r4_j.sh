#!/bin/bash
# round 4, call j: muse.EMAModel (muse_ema_multi) against the reference golden / the oracle, around a training step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ema.py -q -x 2>&1 | tail -25 > gpurun_out/r4_j_pytest.txt
cat gpurun_out/r4_j_pytest.txt
timeout 300 python scripts/exp/ema_bandwidth.py 2>&1 | grep -v amdgpu.ids | tail -5 > gpurun_out/r4_j_ema_bandwidth.txt
cat gpurun_out/r4_j_ema_bandwidth.txt
