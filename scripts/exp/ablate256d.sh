#!/bin/bash
# DMA issue schedule of the 256 x 256 GEMM (round 6): the product kernel spreads a tile's 8 pieces over groups 13-15 of tile t and 0-4 of tile t + 1
# (the late ones have ~1250 cycles to land before the barrier); variant 1 issues all eight in groups 13-15, variant 2 two per group in 13-15 and 0.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I open-muse_amd/csrc scripts/exp/gemm256g.hip"
hipcc $F -o /tmp/g0 & hipcc $F -DG256_SPREAD_EARLY=1 -o /tmp/g1 & hipcc $F -DG256_SPREAD_EARLY=2 -o /tmp/g2 & wait
for v in 1 2; do echo "== correctness variant $v"; timeout 120 /tmp/g$v q | grep -c "OK$"; done
for rep in 1 2; do
echo "== product schedule"; timeout 60 /tmp/g0 t
echo "== all eight pieces behind the barrier"; timeout 60 /tmp/g1 t
echo "== two per group"; timeout 60 /tmp/g2 t
done
