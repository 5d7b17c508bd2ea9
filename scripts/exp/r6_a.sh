#!/bin/bash
# round 6, first GPU call: persistent GN-fused convolution (parity, time, timeline) + the GEMM start-stagger retest
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
T=$O/r06_conv_persist.txt; : > $T
for g in 3 8; do
  echo "=== parity: MUSE_CONV_PERSIST=1 MIN=0 GRID=$g" >> $T
  MUSE_CONV_PERSIST=1 MUSE_CONV_PERSIST_MIN=0 MUSE_CONV_PERSIST_GRID=$g timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -p no:cacheprovider -k "conv_with_fused_groupnorm" 2>&1 | tail -5 >> $T
done
for p in 0 1; do
  echo "=== time: MUSE_CONV_PERSIST=$p" >> $T
  MUSE_CONV_PERSIST=$p timeout 300 python scripts/exp/conv_gn_time.py 2>&1 | grep -v amdgpu.ids >> $T
done
for p in 0 1; do
  echo "=== timeline 128^2: MUSE_CONV_PERSIST=$p" >> $T
  MUSE_HIP_LIB=$PWD/open-muse_amd/csrc/variants/libmuse_hip_ts.so MUSE_CONV_PERSIST=$p timeout 300 python scripts/exp/conv_ts2.py 128 2>&1 | grep -v amdgpu.ids >> $T
  echo "=== timeline 256^2: MUSE_CONV_PERSIST=$p" >> $T
  MUSE_HIP_LIB=$PWD/open-muse_amd/csrc/variants/libmuse_hip_ts.so MUSE_CONV_PERSIST=$p timeout 300 python scripts/exp/conv_ts2.py 256 2>&1 | grep -v amdgpu.ids >> $T
done
cat $T
G=$O/r06_g256p_stagger.txt; : > $G
export MUSE_GEMM256=1
for s in 1 4 8 1; do
  MUSE_G256P_STAGGER=$s timeout 300 python scripts/exp/g256p_probe.py time 2>&1 | grep -v amdgpu.ids | sed "s/^\[/[stagger=$s /" >> $G
done
cat $G
