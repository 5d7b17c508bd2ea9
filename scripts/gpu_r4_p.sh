#!/bin/bash
# round 4, call p: every constructible reference configuration at full size: two optimisation steps each
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_reference_configs.py -q -s 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|float(loss)" | tail -40 > gpurun_out/r4_p_pytest.txt
cat gpurun_out/r4_p_pytest.txt
