#!/bin/bash
# round 4, call q: save_pretrained -> from_pretrained -> compute for every model class
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_models.py -q -x -k "after_save_and_from_pretrained or pipeline_class_conditional" 2>&1 | grep -v "amdgpu.ids" | tail -25 > gpurun_out/r4_q_pytest.txt
cat gpurun_out/r4_q_pytest.txt
