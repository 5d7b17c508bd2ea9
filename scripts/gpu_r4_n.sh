#!/bin/bash
# round 4, call n: the train_muse.py loop body end to end against the oracles
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train_muse_loop.py -q -x 2>&1 | grep -v amdgpu.ids | tail -25 > gpurun_out/r4_n_pytest.txt
cat gpurun_out/r4_n_pytest.txt
