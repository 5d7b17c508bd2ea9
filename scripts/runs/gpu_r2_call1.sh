#!/bin/bash
# round 2, call 1: the two never-executed U-ViT paths, U-ViT bf16 timing + rocprof ranking of its kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
MUSE_TEST_UNVERIFIED=1 timeout 300 python -m pytest tests/test_gpu_uvit.py -q --tb=short -p no:cacheprovider -k "fused_adamw or bf16_mode" > $O/r2_unverified.txt 2>&1
echo "exit $?" >> $O/r2_unverified.txt; tail -30 $O/r2_unverified.txt
timeout 200 python scripts/uvit_bench.py 8 2 f32 > $O/r2_uvit_f32.txt 2>&1; tail -2 $O/r2_uvit_f32.txt
timeout 200 python scripts/uvit_bench.py 8 2 bf16 > $O/r2_uvit_bf16.txt 2>&1; tail -2 $O/r2_uvit_bf16.txt
rm -rf $O/prof_uvit
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_uvit -o uv -- python scripts/uvit_bench.py 8 2 bf16 > $O/r2_uvit_prof.txt 2>&1
f=$(find $O/prof_uvit -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r2_uvit_bf16_kernel_stats.csv && head -30 "$f" | cut -c1-200
find $O/prof_uvit -name "*kernel_trace*" -delete
