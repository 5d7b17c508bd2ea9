#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for v in "" NO_MFMA NO_DMA; do
  echo "== ${v:-full}"
  if [ -n "$v" ]; then export MUSE_HIP_LIB=$PWD/scripts/exp/lib/libmuse_$v.so; fi
  WHICH=conv timeout 200 python scripts/gemm_probe.py 2>&1 | grep "DMA"
done
