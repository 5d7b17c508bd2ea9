#!/bin/bash
# round 6: (1) residual layers, base vs the new launch-per-tile kernel, same box; (2) persistent convolution in k-tile workgroups on the step
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
T=$O/r06_conv_e.txt; : > $T
V=$PWD/open-muse_amd/csrc/variants
for rep in 1 2; do for n in base il0; do
  echo "=== $n rep $rep" >> $T
  MUSE_HIP_LIB=$V/libmuse_hip_$n.so timeout 300 python scripts/exp/conv_ts2.py 256 2>&1 | grep "per launch" >> $T
  MUSE_HIP_LIB=$V/libmuse_hip_$n.so timeout 300 python scripts/exp/conv_ts2.py 128 2>&1 | grep "per launch" >> $T
done; done
cat $T
bash scripts/gpu.sh ab r06_conv_e "MUSE_CONV_PERSIST=0" "MUSE_CONV_PERSIST=1 MUSE_CONV_PERSIST_TILES=2" "MUSE_CONV_PERSIST=1 MUSE_CONV_PERSIST_TILES=4" "MUSE_CONV_PERSIST=1 MUSE_CONV_PERSIST_TILES=8" "MUSE_CONV_PERSIST=0"
