#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for s in 1 2; do MUSE_CONV_SLAB=$s timeout 200 python scripts/exp/conv_seam.py 2>&1 | grep -v amdgpu.ids; done
