#!/bin/bash
# round 6: persistent convolution as the default where it runs alone: tokenizer / decoder tests, then config 5 and the step legs with MUSE_CONV_PERSIST=0 / default
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
T=$O/r06_persist_default.txt; : > $T
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "vqgan or vq_ or conv or taming or decode or smoke or train_step or prefetch" 2>&1 | grep -E "passed|failed" | tail -3 >> $T
for v in 0 1 0 1; do
  echo "MUSE_CONV_PERSIST=$v config 5 (encode -> decode), taming" >> $T
  MUSE_CONV_PERSIST=$v timeout 600 python bench.py --leg vqgan,64 2>/dev/null | tail -1 | cut -c1-260 >> $T
done
for v in 0 1; do
  echo "MUSE_CONV_PERSIST=$v inline-tokenizer step / default (prefetch) step" >> $T
  MUSE_CONV_PERSIST=$v timeout 600 python bench.py --leg run,B,bf16x3,inline,10,64 2>/dev/null | tail -1 >> $T
  MUSE_CONV_PERSIST=$v timeout 600 python bench.py --leg run,B,bf16x3,plain,10,64 2>/dev/null | tail -1 >> $T
done
cat $T
