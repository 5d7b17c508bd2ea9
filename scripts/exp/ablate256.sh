#!/bin/bash
# Ablation of the 256 x 256 LDS-DMA GEMM (open-muse_amd/csrc/gemm256.h) in the stand-alone harness: full kernel, without the
# in-loop DMA, without the MFMAs, with neither (fragment reads + barriers + epilogue); then the BK = 32 five-stage pipeline and
# padded leading dimensions (PAD elements) for the L2-channel question.  Timing only ("t"): ablated results are wrong by design.
#   /usr/local/graft/bin/gpurun -- 'bash scripts/exp/ablate256.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I open-muse_amd/csrc scripts/exp/gemm256g.hip"
hipcc $F -o /tmp/g0 & hipcc $F -DG256_ABLATE_NO_DMA -o /tmp/g1 & hipcc $F -DG256_ABLATE_NO_MFMA -o /tmp/g2 & hipcc $F -DG256_ABLATE_NO_DMA -DG256_ABLATE_NO_MFMA -o /tmp/g3 & wait
echo "== correctness (every layout, ragged shapes, split-K)"; timeout 120 /tmp/g0 q | grep -c "OK$"
echo "== full (BK=64, 2 stages)"; timeout 60 /tmp/g0 t
echo "== no DMA in loop"; timeout 60 /tmp/g1 t
echo "== no MFMA"; timeout 60 /tmp/g2 t
echo "== neither"; timeout 60 /tmp/g3 t
echo "== BK=32, 5 stages: full / no MFMA"; MUSE_G256_BK=32 timeout 60 /tmp/g0 t; MUSE_G256_BK=32 timeout 60 /tmp/g2 t
echo "== padded leading dimensions (PAD=64): full / no MFMA"; PAD=64 timeout 60 /tmp/g0 t; PAD=64 timeout 60 /tmp/g2 t
