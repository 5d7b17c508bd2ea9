#!/bin/bash
# Ablation of the LDS-DMA bf16x3 convolution (open-muse_amd/csrc/conv_dma.hip): the full kernel vs builds without the MFMAs
# (-DCDMA_ABLATE_NO_MFMA) and without the in-loop DMA (-DCDMA_ABLATE_NO_DMA); results are wrong by construction, only the
# timing matters.  The variant libraries are built next to the object files of the normal build and selected with MUSE_HIP_LIB.
#   /usr/local/graft/bin/gpurun -- 'bash scripts/exp/ablate_conv.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
C=open-muse_amd/csrc; L=/tmp/muse_ablate; mkdir -p $L
make -C $C > /dev/null
for v in NO_MFMA NO_DMA; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DCDMA_ABLATE_$v -c $C/conv_dma.hip -o $L/conv_dma_$v.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libmuse_$v.so $C/gemm.o $C/rowops.o $C/vqgan.o $C/attention.o $C/conv_split.o $L/conv_dma_$v.o
done
for v in "" NO_MFMA NO_DMA; do
  echo "== ${v:-full}"
  if [ -n "$v" ]; then export MUSE_HIP_LIB=$L/libmuse_$v.so; fi
  WHICH=conv timeout 200 python scripts/gemm_probe.py 2>&1 | grep "DMA"
done
