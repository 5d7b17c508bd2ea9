#!/bin/bash
# Build gemm.hip with -DG256_TIMESTAMPS next to the normal objects and print the phase timeline of the 256^2 GEMM (scripts/exp/gemm_ts.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
C=open-muse_amd/csrc; L=/tmp/muse_ts; mkdir -p $L
make -C $C > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DG256_TIMESTAMPS -c $C/gemm.hip -o $L/gemm_ts.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libmuse_gts.so $L/gemm_ts.o $C/rowops.o $C/vqgan.o $C/attention.o $C/conv_split.o $C/conv_dma.o $C/uvit.o $C/sampling.o $C/embed.o
MUSE_HIP_LIB=$L/libmuse_gts.so timeout 200 python scripts/exp/gemm_ts.py 2>&1 | grep -v amdgpu.ids
