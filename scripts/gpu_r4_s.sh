#!/bin/bash
# round 4, call s: the 512-pixel research configuration at its yaml batch size (64) next to 32, and the non-downsampled form
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
timeout 600 python scripts/exp/research512_step.py 32 4 2>&1 | grep "research_run"
timeout 600 python scripts/exp/research512_step.py 64 4 2>&1 | grep "research_run"
} > gpurun_out/r4_s_research512.txt
cat gpurun_out/r4_s_research512.txt
