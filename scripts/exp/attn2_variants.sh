#!/bin/bash
# Build variant libraries of attention2.hip next to the product library (run HERE: hipcc cross-compiles; the .so files travel with
# the gpurun snapshot; muse/_hip.py loads whatever MUSE_HIP_LIB names).  Usage: scripts/exp/attn2_variants.sh
set -e
cd "$(dirname "$0")/../../open-muse_amd/csrc"
make -j4 > /dev/null
mkdir -p variants
OBJS="gemm.o gemm_p.o rowops.o vqgan.o attention.o conv_split.o conv_dma.o uvit.o sampling.o embed.o"
build () { # name, extra flags
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $2 -c attention2.hip -o variants/attention2_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libmuse_hip_$1.so $OBJS variants/attention2_$1.o
  echo "built variants/libmuse_hip_$1.so ($2)"
}
build u0 "-DATT2_BWD_UNROLL=0"
build u1 "-DATT2_BWD_UNROLL=1"
build noslp "-fno-slp-vectorize"
build u0noslp "-DATT2_BWD_UNROLL=0 -fno-slp-vectorize"
build ts "-DATT2_TS"
