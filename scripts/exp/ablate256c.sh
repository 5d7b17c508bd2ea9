#!/bin/bash
# L2 touch-prefetch experiment on the 256 x 256 GEMM (round 6): every lane loads 4 bytes of one 128-byte operand line D K-tiles ahead, so that
# the LDS-DMA of that K-tile hits L2.  Correctness (k-contiguous layouts), then time for D = 3, 4, 6 against the plain kernel.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I open-muse_amd/csrc scripts/exp/gemm256g.hip"
hipcc $F -o /tmp/g0 & hipcc $F -DG256_L2_TOUCH=3 -o /tmp/g3 & hipcc $F -DG256_L2_TOUCH=4 -o /tmp/g4 & hipcc $F -DG256_L2_TOUCH=6 -o /tmp/g6 & wait
echo "== correctness with the touch (D = 4)"; timeout 120 /tmp/g4 q | grep -c "OK$"; timeout 120 /tmp/g4 q | grep -v "OK$" | head -5
for rep in 1 2; do
echo "== plain"; timeout 60 /tmp/g0 t
for d in 3 4 6; do echo "== touch $d K-tiles ahead"; timeout 60 /tmp/g$d t; done
done
