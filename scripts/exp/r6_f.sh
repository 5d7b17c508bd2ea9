#!/bin/bash
# round 6: vector-memory path microbenchmark with the GEMM's DMA + store mix; config 4 at seq 1024 in the bf16x3 mode (first run of that leg)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 scripts/exp/vmem_path.hip -o /tmp/vmem_path && timeout 300 /tmp/vmem_path > $O/r06_vmem_path2.txt 2>&1
grep -E "mix|DMA" $O/r06_vmem_path2.txt | tail -20
timeout 900 python bench.py --uvit-leg 32,1024,2,x3 > $O/r06_c4_seq1024_x3.txt 2>&1; tail -3 $O/r06_c4_seq1024_x3.txt | cut -c1-600
timeout 900 python bench.py --uvit-leg 48,1024,2,x3 > $O/r06_c4_seq1024_x3_b48.txt 2>&1; tail -3 $O/r06_c4_seq1024_x3_b48.txt | cut -c1-400
