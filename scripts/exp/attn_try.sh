#!/bin/bash
# attention kernel experiment: parity tests of the fused attention + the micro-benchmark
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention" 2>&1 | tail -4
python scripts/attn_bench.py 50 2>&1 | grep -v amdgpu
} > gpurun_out/attn_try.txt
cat gpurun_out/attn_try.txt
