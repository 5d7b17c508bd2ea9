#!/bin/bash
# Build the timestamped variant of conv_dma.hip (-DCDMA_TIMESTAMPS) next to the normal objects and print the per-tile phase timeline
# of both patch-slab kernels (scripts/exp/conv_ts.py).   /usr/local/graft/bin/gpurun -- 'bash scripts/exp/conv_ts.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
C=open-muse_amd/csrc; L=/tmp/muse_ts; mkdir -p $L
make -C $C > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DCDMA_TIMESTAMPS -c $C/conv_dma.hip -o $L/conv_dma_ts.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libmuse_ts.so $C/gemm.o $C/rowops.o $C/vqgan.o $C/attention.o $C/conv_split.o $L/conv_dma_ts.o $C/uvit.o $C/sampling.o
for s in 1 2; do echo "=== MUSE_CONV_SLAB=$s"; MUSE_HIP_LIB=$L/libmuse_ts.so MUSE_CONV_SLAB=$s timeout 200 python scripts/exp/conv_ts.py 2>&1 | grep -v "amdgpu.ids\|first rows"; done
