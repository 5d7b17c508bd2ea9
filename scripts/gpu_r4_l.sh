#!/bin/bash
# round 4, call l: one-off step measurement of the 512-pixel research configuration (force_down_up_sample + EMA)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python scripts/exp/research512_step.py 32 4 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r4_l_research512.txt
cat gpurun_out/r4_l_research512.txt
