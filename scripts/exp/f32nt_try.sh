#!/bin/bash
# persistent 256^2 kernel for f32 output with k-major B (MUSE_G256P_F32NT): kernel test + config-4 leg A/B on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "persistent_matches" 2>&1 | tail -3
for v in 0 1 0 1; do echo "MUSE_G256P_F32NT=$v"; MUSE_G256P_F32NT=$v timeout 300 python bench.py --uvit-leg 128,256,4 2>/dev/null | tail -1; done
for v in 0 1; do echo "seq1024 MUSE_G256P_F32NT=$v"; MUSE_G256P_F32NT=$v timeout 300 python bench.py --uvit-leg 48,1024,3 2>/dev/null | tail -1; done
} > gpurun_out/f32nt_try.txt 2>&1
cat gpurun_out/f32nt_try.txt
