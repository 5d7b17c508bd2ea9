#!/bin/bash
# conv_seam.py on the normal library and on a build whose epilogue stores nothing (-DCDMA_ABLATE_NO_STORE, built by the caller as
# open-muse_amd/muse/libmuse_hip_nostore.so): how much of the per-tile seam is the output store
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for v in full nostore; do
  if [ $v == nostore ]; then export MUSE_HIP_LIB=$PWD/open-muse_amd/muse/libmuse_hip_nostore.so; else unset MUSE_HIP_LIB; fi
  echo "=== $v"; MUSE_CONV_SLAB=1 timeout 200 python scripts/exp/conv_seam.py 2>&1 | grep -E "Cin  128|Cin  512|per tile"
done
