#!/bin/bash
# round 4, call m: the taming tokenizer on a 512 x 512 picture against the oracle
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_models.py -q -x -s -k "taming_vqgan_f16_8192" 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/r4_m_pytest.txt
cat gpurun_out/r4_m_pytest.txt
