#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for s in 1 2; do for g in 0 1 2; do echo "--- stagger $g"; MUSE_CONV_STAGGER=$g MUSE_CONV_SLAB=$s timeout 200 python scripts/exp/conv_seam.py 2>&1 | grep -E "SLAB|Cin  128|Cin  512|per tile"; done; done
MUSE_CONV_SLAB=2 timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "conv2d" 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "conv2d" 2>&1 | tail -2
