#!/bin/bash
# round 6: weight operand planes kept across steps and refreshed by FusedAdamW (bf16x3 mode): tests, then the config-4 legs with / without
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
T=$O/r06_weight_planes.txt; : > $T
timeout 1500 python -m pytest tests/test_gpu_uvit.py -q -m gpu -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 >> $T
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "adamw or x3 or bf16x3" 2>&1 | grep -E "passed|failed" | tail -3 >> $T
for v in 1 0 1; do
  echo "MUSE_X3_WEIGHT_PLANES=$v" >> $T
  MUSE_X3_WEIGHT_PLANES=$v timeout 900 python bench.py --uvit-leg 64,256,3,x3 2>/dev/null | tail -1 | cut -c1-200 >> $T
done
MUSE_X3_WEIGHT_PLANES=1 timeout 900 python bench.py --uvit-leg 32,1024,2,x3 2>/dev/null | tail -1 | cut -c1-200 >> $T
cat $T
