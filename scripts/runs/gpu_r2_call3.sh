#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for m in 1 2; do
  export MUSE_HIP_LIB=$PWD/open-muse_amd/muse/libmuse_tail$m.so
  echo "== tail mode $m" > $O/r2_tail$m.txt
  timeout 600 python -m pytest tests/test_gpu_kernels.py -q --tb=line -p no:cacheprovider -k "attention" >> $O/r2_tail$m.txt 2>&1
  timeout 300 python scripts/attn_bench.py 20 2>&1 | head -2 >> $O/r2_tail$m.txt
  tail -8 $O/r2_tail$m.txt | cut -c1-200
done
