#!/bin/bash
# Build variant libraries of conv_dma.hip next to the normal objects (they travel with the gpurun snapshot; muse/_hip.py loads MUSE_HIP_LIB).
#   scripts/exp/conv_variants.sh name "flags" [source]      e.g.  conv_variants.sh il0 "-DGN_XF_INTERLEAVE=0"
cd "$(dirname "$0")/../../open-muse_amd/csrc"; mkdir -p variants
OBJS="gemm.o gemm_p.o rowops.o vqgan.o attention.o attention2.o attention3.o conv_split.o uvit.o sampling.o embed.o"
src=${3:-conv_dma.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $2 -I. -c $src -o variants/conv_dma_$1.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libmuse_hip_$1.so $OBJS variants/conv_dma_$1.o && echo "built variants/libmuse_hip_$1.so ($2 $src)"
