#!/bin/bash
# round 6: persistent GN-fused convolution - parity (all cases), then the same-box A/B of the headline step
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
T=$O/r06_conv_persist_b.txt; : > $T
for g in 3 8; do
  echo "=== parity: MUSE_CONV_PERSIST=1 MIN=0 GRID=$g" >> $T
  MUSE_CONV_PERSIST=1 MUSE_CONV_PERSIST_MIN=0 MUSE_CONV_PERSIST_GRID=$g timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "conv_with_fused_groupnorm" 2>&1 | tail -8 >> $T
done
echo "=== tokenizer parity with the persistent kernel" >> $T
MUSE_CONV_PERSIST=1 timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -p no:cacheprovider -k "vqgan_f16_256_vs_reference_golden or vq_indices_over_bench_batch" 2>&1 | tail -6 >> $T
cat $T
bash scripts/gpu.sh ab r06_persist "MUSE_CONV_PERSIST=0" "MUSE_CONV_PERSIST=1" "MUSE_CONV_PERSIST=1 MUSE_CONV_PERSIST_GRID=240" "MUSE_CONV_PERSIST=1 MUSE_CONV_PERSIST_GRID=224" "MUSE_CONV_PERSIST=0"
