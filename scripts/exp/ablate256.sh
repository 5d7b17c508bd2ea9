#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I open-muse_amd/csrc scripts/exp/gemm256g.hip"
hipcc $F -o /tmp/g0 & hipcc $F -DG256_ABLATE_NO_MFMA -o /tmp/g2 & wait
export MUSE_G256_BK=64
for pad in 0 64 8; do echo "== PAD=$pad full"; PAD=$pad timeout 60 /tmp/g0 t; echo "== PAD=$pad no MFMA"; PAD=$pad timeout 60 /tmp/g2 t; done
PAD=64 timeout 60 /tmp/g0 q | grep -c OK
