#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_uvit.py tests/test_gpu_kernels.py -q --tb=short -p no:cacheprovider -k "uvit or attention" > $O/r2_uvit_tests.txt 2>&1
tail -25 $O/r2_uvit_tests.txt | cut -c1-220
for a in "8 2 bf16 256" "32 2 bf16 256 adamw" "8 2 bf16 1024"; do timeout 300 python scripts/uvit_bench.py $a 2>&1 | tail -1; done | tee $O/r2_uvit_bench.txt
rm -rf $O/prof_uvit
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_uvit -o uv -- python scripts/uvit_bench.py 32 2 bf16 256 adamw > $O/r2_uvit_prof.txt 2>&1
f=$(find $O/prof_uvit -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r2_uvit_bf16_b32_kernel_stats.csv && head -24 "$f" | cut -c1-180
find $O/prof_uvit -name "*kernel_trace*" -delete
