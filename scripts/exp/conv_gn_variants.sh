#!/bin/bash
# what the fused GroupNorm input costs the patch-slab convolution: libraries with the SiLU and / or the LDS scale-shift reads compiled out
# (build here: bash scripts/exp/conv_gn_variants.sh build; run on the GPU box: bash scripts/exp/conv_gn_variants.sh)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
C=open-muse_amd/csrc
if [ "$1" = build ]; then
  mkdir -p gpurun_out/gnx
  for v in NO_SILU CONST_SS BOTH; do
    f="-DGNX_$v"; [ $v = BOTH ] && f="-DGNX_NO_SILU -DGNX_CONST_SS"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $f -c $C/conv_dma.hip -o /tmp/conv_dma_$v.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/exp/libmuse_gnx_$v.so $C/gemm.o $C/gemm_p.o $C/rowops.o $C/vqgan.o $C/attention.o $C/conv_split.o /tmp/conv_dma_$v.o $C/uvit.o $C/sampling.o $C/embed.o
  done
  exit 0
fi
python scripts/exp/conv_gn_time.py 1 2>&1 | grep -v amdgpu.ids
for v in NO_SILU CONST_SS BOTH; do echo "== $v"; MUSE_HIP_LIB=$PWD/scripts/exp/libmuse_gnx_$v.so python scripts/exp/conv_gn_time.py 1 2>&1 | grep -v amdgpu.ids; done
