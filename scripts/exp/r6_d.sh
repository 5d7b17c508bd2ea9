#!/bin/bash
# round 6: same-box A/B of conv_dma.hip variants (open-muse_amd/csrc/variants/libmuse_hip_<name>.so): parity, per-layer time, headline step
#   scripts/exp/r6_d.sh <tag> name1 name2 ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
tag=$1; shift
T=$O/${tag}.txt; : > $T
V=$PWD/open-muse_amd/csrc/variants
for n in "$@"; do
  echo "=== $n: parity" >> $T
  MUSE_HIP_LIB=$V/libmuse_hip_$n.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "conv_with_fused_groupnorm or conv_dma or split2" 2>&1 | tail -3 >> $T
done
for rep in 1 2; do for n in "$@"; do
  echo "=== $n: time (rep $rep)" >> $T
  MUSE_HIP_LIB=$V/libmuse_hip_$n.so timeout 300 python scripts/exp/conv_gn_time.py 3 2>&1 | grep -v amdgpu.ids >> $T
done; done
cat $T
args=()
for n in "$@"; do args+=("MUSE_HIP_LIB=$V/libmuse_hip_$n.so"); done
bash scripts/gpu.sh ab $tag "${args[@]}" "MUSE_HIP_LIB=$V/libmuse_hip_$1.so"
