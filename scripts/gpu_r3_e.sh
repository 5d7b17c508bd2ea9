#!/bin/bash
# round 3, run E: persistent patch-slab convolution (MUSE_CONV_SLAB=2) under the token prefetch, with and without a CU-masked
# tokenizer stream (the persistent kernel holds its CUs for a whole launch: alone it starved the step's kernels in round 2)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/exp/libstream_placement.so scripts/exp/stream_placement.hip || exit 1
rm -f $O/r3e_conv_persist.txt
for slab in 1 2; do
  echo "MUSE_CONV_SLAB=$slab" | tee -a $O/r3e_conv_persist.txt
  MUSE_CONV_SLAB=$slab timeout 400 python scripts/exp/stream_placement.py base mask:64x:p mask:128x:p mask:96c:p 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $O/r3e_conv_persist.txt
done
