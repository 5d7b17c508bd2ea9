#!/bin/bash
# What the B operand's LDS-DMA costs the 256 x 256 GEMM's K loop (round 6): full kernel; B pieces issued out of range (zero-fill: the LDS
# write happens, no L2 / memory traffic); B pieces not issued at all; and the existing no-DMA / no-MFMA forms.  Timing only.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I open-muse_amd/csrc scripts/exp/gemm256g.hip"
hipcc $F -o /tmp/g0 & hipcc $F -DG256_ABLATE_B_OOB -o /tmp/g1 & hipcc $F -DG256_ABLATE_NO_B -o /tmp/g2 & hipcc $F -DG256_ABLATE_NO_DMA -o /tmp/g3 & wait
for rep in 1 2; do
echo "== full"; timeout 60 /tmp/g0 t
echo "== B pieces zero-filled (LDS writes, no memory traffic)"; timeout 60 /tmp/g1 t
echo "== B pieces not issued"; timeout 60 /tmp/g2 t
echo "== no DMA at all"; timeout 60 /tmp/g3 t
done
