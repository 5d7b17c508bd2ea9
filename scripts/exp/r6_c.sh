#!/bin/bash
# round 6: launch-per-tile GN-fused convolution with one prologue round trip, bias through LDS, residual prefetched in the last tap
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
T=$O/${1:-r06_conv_c}.txt; : > $T
echo "=== parity" >> $T
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "conv" 2>&1 | tail -6 >> $T
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -p no:cacheprovider -k "vqgan or vq_" 2>&1 | tail -6 >> $T
echo "=== time" >> $T
timeout 300 python scripts/exp/conv_gn_time.py 2>&1 | grep -v amdgpu.ids >> $T
for hw in 128 256; do
  echo "=== timeline $hw^2" >> $T
  MUSE_HIP_LIB=$PWD/open-muse_amd/csrc/variants/libmuse_hip_ts.so timeout 300 python scripts/exp/conv_ts2.py $hw 2>&1 | grep -v amdgpu.ids >> $T
done
cat $T
bash scripts/gpu.sh ab ${1:-r06_conv_c} "MUSE_CONV_PERSIST=0" "MUSE_CONV_PERSIST=0"
