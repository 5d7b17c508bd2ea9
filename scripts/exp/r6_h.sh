#!/bin/bash
# round 6: block-by-block bf16x3 attention (1024-token sequences): kernel tests, the U-ViT tests, then config 4 at seq 1024 in the bf16x3 mode
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
T=$O/r06_attn_blocks.txt; : > $T
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -rP -k "bf16x3" 2>&1 | grep -E "bf16x3 attention|passed|failed|Error|error" | tail -30 >> $T
timeout 1200 python -m pytest tests/test_gpu_uvit.py -q -m gpu -p no:cacheprovider -x 2>&1 | grep -E "passed|failed" | tail -3 >> $T
cat $T
timeout 900 python bench.py --uvit-leg 32,1024,2,x3 > $O/r06_c4_seq1024_x3_blocks.txt 2>&1; tail -1 $O/r06_c4_seq1024_x3_blocks.txt | cut -c1-330
timeout 900 python bench.py --uvit-leg 64,256,2,x3 > $O/r06_c4_seq256_x3.txt 2>&1; tail -1 $O/r06_c4_seq256_x3.txt | cut -c1-330
